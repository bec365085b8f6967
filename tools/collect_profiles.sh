#!/bin/bash
# Runs on the GPU box: bench line + rocprofv3 kernel-trace stats + PMC passes (separate runs, no trace domains mixed
set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/r01
rm -rf $out; mkdir -p $out
BENCH="bench.py --steps 64 --warmup 8 --no-cpu-baseline"
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $BENCH > $out/trace.log 2>&1
python3 tools/prof_summary.py $out amq > $out/summary.txt 2>&1
cat $out/bench.json
tail -20 $out/summary.txt
