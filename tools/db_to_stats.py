"""rocprofv3 results .db (top_kernels view) -> the kernel-stats CSV kept under profiles/ (times in microseconds)."""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels "
                  "order by total_duration desc").fetchall()
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for name, calls, tot, avg, pct in rows:
        w.writerow([name, calls, "%.3f" % tot, "%.3f" % avg, "%.4f" % pct])   # the view reports microseconds
print("%d kernels, %.2f ms" % (len(rows), sum(r[2] for r in rows) / 1e3))
