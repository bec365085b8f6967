#!/bin/bash
for tag in "" lb6 lb8; do
  for depth in 2 4; do
    echo "== lib=${tag:-product} depth=$depth"
    AMQ_LIB_TAG=$tag timeout 200 python tools/microbench.py --iters 100 --gemm 0 --depth $depth 2>&1 | grep -v amdgpu | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if d['M'] == 1 and (d['N'], d['K']) in ((4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008)): print(d['N'], d['K'], d['bits'], d['us'])
" | paste - - - - - - - - - - - -
  done
done
