#!/bin/bash
# usage: [VARIANT=<tag>] tools/attic/mb_short.sh [microbench args]  -> compact per-shape table (us per launch under graph replay)
timeout -k 10 300 python tools/with_variant.py "${VARIANT-product}" tools/microbench.py --iters 200 --gemm 0 "$@" 2>&1 | grep "^{" | python3 -c "
import sys, json
rows = {}
for l in sys.stdin:
    d = json.loads(l)
    if d['M'] != 1: continue
    rows.setdefault((d['N'], d['K']), {})[d['bits']] = (d['us'], d['GBps'])
for (n, k), v in rows.items():
    print('%6d x %6d  ' % (n, k) + '  '.join('b%d %6.2f us %5.0f GB/s' % (b, v[b][0], v[b][1]) for b in (4, 3, 2)))
"
