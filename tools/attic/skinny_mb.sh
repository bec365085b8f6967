#!/bin/bash
# skinny-vs-tiled GEMM crossover: us per launch for a row sweep, forced kernel family 1 (tiled + split-K) vs 2 (skinny, <= 64 rows)
for shape in 4096,4096 11008,4096 4096,11008; do
  for route in 1 2; do
    echo "== N,K = $shape  route = $route"
    bash tools/attic/gemm_mb.sh --route $route --gemm_shape $shape --gemm_m 16,32,64 --gemm_bits 3 --iters 400 || exit 1
  done
done
