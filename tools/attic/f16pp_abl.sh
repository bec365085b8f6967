#!/bin/bash
# timing-only ablations of amq::gemm_f16_pp_kernel (make -C amq_amd/csrc tuvariant TU=amq_gemm_f16 TAG=<t> EXTRA=-DAMQ_PP_ABL_<...>): usage tools/attic/f16pp_abl.sh tag [tag ...]
for t in "$@"; do
  echo "== $t"
  timeout -k 10 200 python tools/with_variant.py $t tools/f16pp_bench.py --no-check --rounds 5 --m 32768 --shapes "13824,5120" --routes dense,torch 2>&1 | grep '"N"' | python3 -c "import sys,json; [print({k:v for k,v in json.loads(l).items() if k.startswith('TF') or k.startswith('us')}) for l in sys.stdin]"
done
