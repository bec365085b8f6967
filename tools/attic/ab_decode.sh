#!/bin/bash
# decode tokens/s of the bench workload for several library builds on ONE box (make tuvariant ...; TAGS="product tagA tagB")
for round in 1 2; do
for tag in ${TAGS-product}; do
  timeout -k 10 200 python tools/with_variant.py $tag bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-layer-table --no-mfma 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$tag', round(d['value'], 1), 'tokens/s  frac', round(d['roofline']['frac'], 4), ' us/launch', round(d['roofline']['us_per_launch'], 3))" || exit 1
done
done
