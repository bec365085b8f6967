#!/bin/bash
# ring-GEMM A/B builds (make tuvariant ...): one process per build, route 3 only.  TAGS="product nobarnox ..."
for tag in ${TAGS-product nobarnox nodeq noldsx nodeqlds prio1 prio2}; do
  echo "== build: $tag"
  timeout -k 10 120 python tools/with_variant.py $tag tools/attic/gemm_routes.py --shapes "${SHAPES-13824,5120}" --m 32768 --bits ${BITS-4,3} --routes 3 --rounds 7 2>&1 | grep "^{" || exit 1
done
