#!/usr/bin/env python3
"""Kernel resource usage of one .hip file: python3 scratch/kres.py <file.hip> [name filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-Iinclude", "-Itransformer_engine_amd/csrc",
       "-fno-gpu-rdc", "-c", src, "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
def flush():
    if cur and flt in cur.get("name", ""):
        dem = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::|te::", "", dem).split("(")[0]
        print(f"{dem[:70]:70s} vgpr {cur.get('VGPRs','?'):>4} agpr {cur.get('AGPRs','?'):>3} spill {cur.get('VGPRs Spill','?'):>4} occ {cur.get('Occupancy [waves/SIMD]','?')} lds {cur.get('LDS Size [bytes/block]','?')}")
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?):\s+(\S+)\s+\[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        flush(); cur = {"name": v}
    else:
        cur[k] = v
flush()
if "error" in out: print(out[-3000:])
