#!/bin/bash
# Runs on the GPU box: the round's bench lines, the rocprofv3 kernel-trace summary of the same bench command, and the PMC
# passes (HBM traffic of the decode GEMV launches: FETCH_SIZE and WRITE_SIZE in separate runs, no trace domains mixed in).
# (before calling it through gpurun, delete the local gpurun_out/<round>: gpurun MERGES, so files of an earlier call would be averaged in)
# usage: tools/collect_round.sh r02   -> gpurun_out/<round>/...; tools/make_profile_summary.py <round> turns it into profiles/<round>_*
set -u
R=${1:-r02}
export TMPDIR=/tmp
out=$PWD/gpurun_out/$R
rm -rf $out; mkdir -p $out
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
timeout -k 10 600 python3 bench.py --config 4 > $out/bench_config4.json 2> $out/bench_config4.err || exit 1
BENCH="bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-layer-table --no-mfma"
cd /tmp && cd - > /dev/null
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $BENCH > $out/trace.log 2>&1 || exit 1
CMD="tools/prof_decode.py 4"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $CMD > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $CMD > $out/pmc_write.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 $CMD > $out/pmc_sq.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq_gs -- python3 $CMD gs > $out/pmc_sq_gs.log 2>&1 || exit 1
python3 tools/make_profile_summary.py $R $out/profiles
cut -c1-600 $out/bench.json; echo; cut -c1-900 $out/bench_config4.json
