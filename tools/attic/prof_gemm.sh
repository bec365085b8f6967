#!/bin/bash
# usage: tools/attic/prof_gemm.sh <tag> [prof_gemm.py args]   -> gpurun_out/prof_<tag>/summary.txt
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/attic/prof_gemm.py "$@" > $out/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $out/pmc1 -- python3 tools/attic/prof_gemm.py "$@" > $out/pmc1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d $out/pmc2 -- python3 tools/attic/prof_gemm.py "$@" > $out/pmc2.log 2>&1 || exit 1
python3 tools/attic/prof_summary.py $out "${FILTER-gemm}" > $out/summary.txt 2>&1
cat $out/summary.txt
