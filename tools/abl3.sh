#!/bin/bash
for w in 4 8 16; do
  for depth in 2 4; do
    echo "== waves=$w depth=$depth"
    timeout 200 python tools/microbench.py --iters 100 --gemm 0 --waves $w --depth $depth 2>&1 | grep -v amdgpu | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['M'] == 1 and (d['N'], d['K']) in ((4096, 4096), (4096, 11008)): print(d['N'], d['K'], d['bits'], d['us'])
" | paste - - - - - -
  done
done
