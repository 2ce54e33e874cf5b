for P in 1 2 4 8; do python scripts/r2_probe_feed.py $P 16 2>&1 | tail -1; done
PBSGPU_TRACE=1 python scripts/r2_probe_feed.py 8 8 > /tmp/tr.log 2>&1; grep "^P=" /tmp/tr.log
grep -o "cut [0-9.]* ms" /tmp/tr.log | awk '{v=$2; if (v<2) a++; else if (v<10) b++; else if (v<100) c++; else d++; s+=v} END {print "cut: <2ms",a," 2-10",b," 10-100",c," >100",d," total_ms",s}'
grep -o "readback [0-9.]* ms" /tmp/tr.log | awk '{v=$2; if (v<2) a++; else if (v<10) b++; else if (v<100) c++; else d++; s+=v} END {print "readback: <2ms",a," 2-10",b," 10-100",c," >100",d," total_ms",s}'
grep -o "append [0-9.]* ms" /tmp/tr.log | awk '{v=$2; if (v<2) a++; else if (v<10) b++; else if (v<100) c++; else d++; s+=v} END {print "append: <2ms",a," 2-10",b," 10-100",c," >100",d," total_ms",s}'
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
