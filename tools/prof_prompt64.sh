#!/bin/bash
# usage: tools/prof_prompt64.sh <tag> [rows]  -> gpurun_out/prof_prompt_<tag>/summary.txt: per-kernel table of the 7B 64-row prompt pass (rocprofv3 kernel trace)
set -u
tag=$1; rows=${2:-64}
out=$PWD/gpurun_out/prof_prompt_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/prompt64_time.py $rows > $out/trace.log 2>&1 || exit 1
python3 tools/prof_kernel_table.py $out/trace > $out/summary.txt
tail -2 $out/trace.log
head -24 $out/summary.txt
