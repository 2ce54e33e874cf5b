for P in 1 2 3 4 6 8; do python scripts/r2_probe_feed.py $P 16 2>&1 | tail -1; done
timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | tail -3
