#!/bin/bash
# usage: tools/attic/prof_run.sh <tag> N K bits [M]   -> gpurun_out/prof_<tag>/{stats,pmc}
set -u
tag=$1; shift
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/attic/prof_gemv.py "$@" > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/pmc1 -- python3 tools/attic/prof_gemv.py "$@" > $out/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- python3 tools/attic/prof_gemv.py "$@" > $out/pmc2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/pmc3 -- python3 tools/attic/prof_gemv.py "$@" > $out/pmc3.log 2>&1
find $out -name "*.csv" | head -20
