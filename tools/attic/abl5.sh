#!/bin/bash
# ablations of the GEMV body on the gate/up and q/k/v shapes (us per launch under graph replay)
for tag in product abl_NOLOAD abl_NOFINISH abl_NOLOAD_NOFINISH; do
  echo "== lib=$tag"
  VARIANT=$tag tools/attic/mb_short.sh --only 22016,4096 || exit 1
  VARIANT=$tag tools/attic/mb_short.sh --only 4096,4096 || exit 1
done
