"""Print the kernel timeline of the last few steps from a rocprofv3 --kernel-trace CSV (start/end/gaps)."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ev = ev[len(ev) - n - skip: len(ev) - skip]
prev = None
for s, e, name in ev:
    gap = (s - prev) / 1000 if prev else 0.0
    print(f"gap {gap:8.2f} us  dur {(e - s) / 1000:8.2f} us  {name[:70]}")
    prev = e
