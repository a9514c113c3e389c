"""Summarise the SQ / GRBM counter passes of profiles/collect_sq.sh (rocpd databases) per GEMM role.
Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the
SIMDs that report (one MFMA 32x32x64 f8f6f4 = 64 cycles on its SIMD); SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles; GRBM_GUI_ACTIVE is the sum over the 8 XCDs, so the effective clock is
GRBM_GUI_ACTIVE / 8 / kernel duration.  MFMA-busy fraction = MFMA cycles the launch needs (2*n*m*k / 131072 MFMAs
x 64 cycles) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the share of the SIMD-cycles of the launch in which the matrix
pipe was executing, independent of how the SQ counter aggregates; the raw counter is reported beside it.
Roles come from the dispatch order of pmc_target.py (fprop, dgrad, wgrad per repetition).
Usage: python3 profiles/parse_sq.py <out.json> <results.db> [<results.db> ...]"""
import json
import sqlite3
import sys

ROLES = ("gemm_fprop", "gemm_dgrad", "gemm_wgrad")
FLOP = 2.0 * 8192 * 14336 * 4096
MFMA_CYC = FLOP / (2.0 * 32 * 32 * 64) * 64      # SIMD-cycles of matrix work in one launch
out = {"_units": __doc__.split("Usage")[0].strip()}
med = lambda a: sorted(a)[len(a) // 2]
for db in sys.argv[2:]:
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    has_time = "start" in cols and "end" in cols
    sel = "dispatch_id, kernel_name, counter_name, value" + (", start, end" if has_time else "")
    per = {}
    order = []
    for row in cur.execute(f"select {sel} from counters_collection where kernel_name like '%gemm_fp8_kernel%' order by dispatch_id"):
        did, name, cname, val = row[:4]
        if did not in per:
            per[did] = {"dur_ns": (row[5] - row[4]) if has_time else None}
            order.append(did)
        per[did][cname] = per[did].get(cname, 0.0) + float(val)
    for i, did in enumerate(order):
        per[did]["role"] = ROLES[i % 3]
    for role in ROLES:
        rows = [per[d] for d in order if per[d]["role"] == role][1:]      # first repetition = warm-up
        if not rows:
            continue
        o = out.setdefault(role, {})
        for c in sorted({k for r in rows for k in r} - {"role", "dur_ns"}):
            o[c] = med([r[c] for r in rows if c in r])
        if rows[0]["dur_ns"]:
            o["duration_us_under_pmc"] = med([r["dur_ns"] for r in rows]) / 1e3
        if "GRBM_GUI_ACTIVE" in o:
            cyc = o["GRBM_GUI_ACTIVE"] / 8.0
            o["gpu_cycles_per_launch"] = cyc
            o["mfma_busy_frac_of_simd_cycles"] = MFMA_CYC / (1024.0 * cyc)
            if o.get("duration_us_under_pmc"):
                o["effective_clock_GHz"] = cyc / (o["duration_us_under_pmc"] * 1e3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o and "SQ_BUSY_CYCLES" in o and o["SQ_BUSY_CYCLES"]:
            o["raw_mfma_busy_over_sq_busy"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["SQ_BUSY_CYCLES"]
        if "SQ_WAVE_CYCLES" in o and o["SQ_WAVE_CYCLES"]:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
                if c in o:
                    o[c + "_share_of_wave_cycles"] = o[c] / o["SQ_WAVE_CYCLES"]
for role in ROLES:
    if role in out:
        print(role, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out[role].items()}))
json.dump(out, open(sys.argv[1], "w"), indent=1)
