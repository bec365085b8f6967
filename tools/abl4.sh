#!/bin/bash
for tag in "" abl_NOMETA abl_NOXLDS abl_NOMETA_NOXLDS; do
    echo "== lib=${tag:-product}"
    AMQ_LIB_TAG=$tag timeout 200 python tools/microbench.py --iters 100 --gemm 0 2>&1 | grep -v amdgpu | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['M'] == 1 and (d['N'], d['K']) in ((4096, 4096), (12288, 4096), (22016, 4096)): print(d['N'], d['K'], d['bits'], d['us'])
" | paste - - - - - - - - -
done
