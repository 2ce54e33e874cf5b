"""Export small CSV summaries from rocprofv3's rocpd sqlite output (this ROCm writes *_results.db by default).

  rocpd_export.py stats    <results.db> <out.csv>     per-kernel calls / total / average / min / max / % (ns)
  rocpd_export.py trace    <results.db> <out.csv>     one row per dispatch: name, queue, stream, start, end (ns)
  rocpd_export.py counters <results.db> <out.csv>     per kernel and counter: launches, mean, min, max of the value
"""
import csv
import sqlite3
import sys


def main():
    mode, db, out = sys.argv[1:4]
    c = sqlite3.connect(db)
    w = csv.writer(open(out, "w", newline=""))
    if mode == "stats":
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
        rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                              "group by name order by sum(duration) desc"))
        tot = sum(r[2] for r in rows) or 1
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), int(r[4]), int(r[5]), round(100.0 * r[2] / tot, 3)])
    elif mode == "trace":
        w.writerow(["Kernel_Name", "Queue_Id", "Stream_Id", "Start_Timestamp", "End_Timestamp", "DurationNs"])
        for r in c.execute("select name, queue_id, stream_id, start, end, duration from kernels order by start"):
            w.writerow(r)
    elif mode == "counters":
        w.writerow(["Kernel_Name", "Counter_Name", "Launches", "Mean", "Min", "Max", "MeanDurationNs"])
        for r in c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) from "
                           "counters_collection group by kernel_name, counter_name order by kernel_name, counter_name"):
            w.writerow([r[0], r[1], r[2], round(r[3], 3), round(r[4], 3), round(r[5], 3), round(r[6], 1)])
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
