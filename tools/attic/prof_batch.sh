#!/bin/bash
# usage: tools/attic/prof_batch.sh <tag> B...   -> gpurun_out/prof_batch_<tag>/B<b>/ (rocprofv3 kernel trace of eager batched decode steps) + summary.txt
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_batch_$tag
mkdir -p $out
export TMPDIR=/tmp
for b in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/B$b -- python3 tools/attic/prof_decode_batch.py $b 8 > $out/B$b.log 2>&1 || exit 1
  echo "== B=$b" >> $out/summary.txt
  python3 tools/prof_kernel_table.py $out/B$b >> $out/summary.txt
done
cat $out/summary.txt
