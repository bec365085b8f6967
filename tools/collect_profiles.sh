#!/bin/bash
# Runs on the GPU box: bench line + rocprofv3 kernel-trace stats + PMC passes (separate runs, no trace domains mixed
# with --pmc) of the SAME bench command; raw output under gpurun_out/r01/, summaries copied to profiles/ afterwards.
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/r01
rm -rf $out; mkdir -p $out
BENCH="bench.py --steps 64 --warmup 8 --no-cpu-baseline"
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $BENCH > $out/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $BENCH > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $BENCH > $out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 $BENCH > $out/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_mfma -- python3 $BENCH > $out/pmc_mfma.log 2>&1
python3 tools/prof_summary.py $out amq > $out/summary.txt 2>&1
cat $out/bench.json
tail -20 $out/summary.txt
