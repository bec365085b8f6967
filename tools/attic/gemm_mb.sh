#!/bin/bash
# GEMM microbench table (5120x5120): TFLOP/s per M and bit-width
timeout -k 10 240 python tools/microbench.py --iters 100 --gemv 0 "$@" 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  M %6d b%d %8.1f us %7.1f TFLOP/s' % (d['M'], d['bits'], d['us'], d['TFLOPs']))
"
