"""Export the per-kernel summary (the --stats view) of a rocprofv3 rocpd database to CSV.
Usage: python3 profiles/export_stats.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["Name", "Calls", "TotalDuration(us)", "AverageDuration(us)", "Percentage"])
    for name, calls, total, avg, pct in rows:
        w.writerow([name[:160], calls, round(total, 3), round(avg, 3), round(pct, 3)])
for r in rows[:8]:
    print(f"{r[0][:90]:90s} calls {r[1]:4d} avg {r[3]:9.2f} us  {r[4]:5.1f}%")
