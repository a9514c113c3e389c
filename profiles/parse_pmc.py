"""Turn the rocpd databases of the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into a
traffic JSON: fabric-side (HBM + Infinity Cache) bytes per launch for each te:: kernel.
Corrections per MI355X_MICROARCH.md 'HBM': both counters are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so it is doubled for the 16-byte-per-lane streaming reads these
kernels issue; WRITE_SIZE is taken as is.
Usage: python3 profiles/parse_pmc.py <fetch_results.db> <write_results.db> <out.json>"""
import json
import re
import sqlite3
import sys


def collect(db, counter):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, grid_size, value from counters_collection where counter_name = ? "
         "and kernel_name like '%te::%' order by dispatch_id")
    per = {}
    for name, grid, value in cur.execute(q, (counter,)):
        m = re.search(r"(\w+_kernel)<([^>]*)>", name)
        key = (f"{m.group(1)}<{m.group(2)}>" if m else name[:60]) + f" grid={grid}"
        per.setdefault(key, []).append(float(value))
    return per


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for key in sorted(set(fetch) | set(write)):
    f = [2.0 * 1024.0 * v for v in fetch.get(key, [])]
    w = [1024.0 * v for v in write.get(key, [])]
    med = lambda a: sorted(a)[len(a) // 2] if a else None
    out[key] = {"launches": max(len(f), len(w)), "fetch_bytes_per_launch_median": med(f),
                "write_bytes_per_launch_median": med(w),
                "traffic_bytes_per_launch": (med(f) or 0) + (med(w) or 0)}
    print(f"{key[:80]:80s} fetch {out[key]['fetch_bytes_per_launch_median'] / 1e6 if f else 0:9.1f} MB  "
          f"write {out[key]['write_bytes_per_launch_median'] / 1e6 if w else 0:9.1f} MB  (n={out[key]['launches']})")
json.dump(out, open(sys.argv[3], "w"), indent=1)
