set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/gap_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap_trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-grouped > gpurun_out/gap_trace.json 2> gpurun_out/gap_trace.err || { tail gpurun_out/gap_trace.err; exit 1; }
python - <<'PY'
import csv, glob, statistics as st
f = glob.glob("gpurun_out/gap_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    if "cast_flat" in n: return "cast"
    if "delayed_scaling" in n: return "recipe"
    if "gemm_fp8_kernel" in n:
        a = n.split("gemm_fp8_kernel<")[1].split(">")[0].replace(" ", "").split(",")
        return "gemm" + a[0] + "km" + a[-1]
    return n[:20]
seq = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
pat = ["cast", "gemm0km0", "cast", "gemm1km2", "gemm1km3", "recipe"]
steps = []
i = 0
while i + 6 <= len(seq):
    if [s[0] for s in seq[i:i+6]] == pat:
        steps.append(seq[i:i+6]); i += 6
    else:
        i += 1
print("steps found", len(steps))
# consecutive steps (graph replay): those whose next step starts within 30 us
tight = [k for k in range(len(steps)-1) if steps[k+1][0][1] - steps[k][5][2] < 30000]
print("tight", len(tight))
import collections
gaps = collections.defaultdict(list); durs = collections.defaultdict(list); tot = []
for k in tight:
    s = steps[k]
    for j in range(5):
        gaps[j].append((s[j+1][1] - s[j][2]) / 1e3)
    gaps[5].append((steps[k+1][0][1] - s[5][2]) / 1e3)
    for j in range(6): durs[j].append((s[j][2] - s[j][1]) / 1e3)
    tot.append((steps[k+1][0][1] - s[0][1]) / 1e3)
print("median step", st.median(tot))
for j in range(6):
    print(pat[j], "dur", round(st.median(durs[j]), 1), "gap after", round(st.median(gaps[j]), 2))
PY
