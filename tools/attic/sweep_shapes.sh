#!/bin/bash
# 7B and 13B decode launch shapes, 3 bit, one row: waves x rpt sweep
for shape in "12288,4096" "22016,4096" "4096,11008" "15360,5120" "27648,5120" "5120,13824"; do
  for w in 0 8 16; do
    for r in 0 1 2 3 4; do
      echo -n "shape $shape waves $w rpt $r: "
      timeout -k 10 60 python tools/microbench.py --gemm 0 --iters 100 --only "$shape" --waves $w --rpt $r 2>&1 | grep '"bits": 3' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['us'], d['GBps'])
"
    done
  done
done
