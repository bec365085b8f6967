#!/bin/bash
# skinny-vs-tiled GEMM crossover: us per launch for a row sweep, AMQ_GEMM_SKINNY_MAX = 0 (tiled + split-K) vs 4096 (skinny)
for shape in 4096,4096 11008,4096 4096,11008; do
  for lim in 0 4096; do
    echo "== N,K = $shape  skinny_max = $lim"
    AMQ_GEMM_SKINNY_MAX=$lim bash tools/gemm_mb.sh --gemm_shape $shape --gemm_m 16,32,64,128,256,512 --gemm_bits 3 --iters 400 || exit 1
  done
done
