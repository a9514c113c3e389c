"""Turn the rocpd databases of the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
profiles/pmc_target.py into a traffic JSON: fabric-side (HBM + Infinity Cache) bytes per launch
for each hot kernel of the configs[1] step.
Corrections per MI355X_MICROARCH.md 'HBM': both counters are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so it is doubled for the 16-byte-per-lane streaming reads these
kernels issue; WRITE_SIZE is taken as is.
Roles come from the dispatch order of pmc_target.py: three set-up cast_transposes (X, W, dY), then
per repetition cast_transpose X, GEMM fprop, cast_transpose dY, GEMM dgrad, GEMM wgrad.
Usage: python3 profiles/parse_pmc.py <fetch_results.db> <write_results.db> <out.json>"""
import json
import sqlite3
import sys

GEMM_ROLES = ("gemm_fprop", "gemm_dgrad", "gemm_wgrad")
CT_SETUP = ("setup_cast_transpose_x", "setup_cast_transpose_w", "setup_cast_transpose_dy")
CT_ROLES = ("cast_transpose_x", "cast_transpose_dy")


def collect(db, counter):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, value from counters_collection where counter_name = ? "
         "and kernel_name like '%te::%' order by dispatch_id")
    per, n_gemm, n_ct = {}, 0, 0
    for name, value in cur.execute(q, (counter,)):
        if "gemm_fp8_kernel" in name:
            role = GEMM_ROLES[n_gemm % 3]
            n_gemm += 1
        elif "cast_transpose_tile_kernel" in name:
            role = CT_SETUP[n_ct] if n_ct < 3 else CT_ROLES[(n_ct - 3) % 2]
            n_ct += 1
        else:
            continue
        per.setdefault(role, []).append(float(value))
    return per


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
med = lambda a: sorted(a)[len(a) // 2] if a else None
out = {}
for key in sorted(set(fetch) | set(write)):
    f = [2.0 * 1024.0 * v for v in fetch.get(key, [])]
    w = [1024.0 * v for v in write.get(key, [])]
    out[key] = {"launches": max(len(f), len(w)), "fetch_bytes_per_launch_median": med(f),
                "write_bytes_per_launch_median": med(w),
                "traffic_bytes_per_launch": (med(f) or 0) + (med(w) or 0)}
    print(f"{key:28s} fetch {(med(f) or 0) / 1e6:9.1f} MB  write {(med(w) or 0) / 1e6:9.1f} MB  (n={out[key]['launches']})")
json.dump(out, open(sys.argv[3], "w"), indent=1)
