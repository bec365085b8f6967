#!/bin/bash
# GEMM ablations: TFLOP/s at 5120x5120 for M in 512..16384
for tag in product gabl_NOLOADA gabl_NODEQ gabl_NOLOADA_NOBAR; do
  echo "== lib=$tag"
  timeout -k 10 200 python tools/with_variant.py $tag tools/microbench.py --iters 100 --gemv 0 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  M %6d b%d %8.1f us %7.1f TFLOP/s' % (d['M'], d['bits'], d['us'], d['TFLOPs']))
" || exit 1
done
