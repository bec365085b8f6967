#!/bin/bash
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/r01
mkdir -p $out
CMD="tools/prof_decode.py 4"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $CMD > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $CMD > $out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 $CMD > $out/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $out/pmc_mfma -- python3 $CMD > $out/pmc_mfma.log 2>&1
python3 tools/prof_summary.py $out amq > $out/summary.txt 2>&1
grep -E "^PMC" $out/summary.txt | cut -c1-400
