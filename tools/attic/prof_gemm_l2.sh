#!/bin/bash
# usage: tools/attic/prof_gemm_l2.sh <tag> [prof_gemm.py args] -> L2 hit / fetch counters of a GEMM launch (separate PMC passes)
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/pmc_l2 -- python3 tools/attic/prof_gemm.py "$@" > $out/pmc_l2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 tools/attic/prof_gemm.py "$@" > $out/pmc_fetch.log 2>&1 || exit 1
python3 tools/attic/prof_summary.py $out "${FILTER-gemm}" > $out/summary.txt 2>&1
cat $out/summary.txt
