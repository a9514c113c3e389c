"""Instruction mix of the MFMA-heaviest basic block (the K-tile loop) of every gemm_fp8_kernel in a device .s file."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
funcs = re.split(r'\n(?=_ZN2te12_GLOBAL__N_115gemm_fp8_kernel\S*:)', s)
for f in funcs[1:1 + int(sys.argv[2]) if len(sys.argv) > 2 else None]:
    name = f.split(':')[0]
    blocks = re.split(r'\n(?=\.LBB[0-9_]+:)', f)
    best = max(blocks, key=lambda b: b.count('v_mfma'))
    ins = [l.strip() for l in best.split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
    c = Counter(i.split()[0] for i in ins)
    print(name[40:], 'blocks', len(blocks), 'loop ins', len(ins), 'mfma', best.count('v_mfma'))
    print('   ', dict(c.most_common(30)))
