#!/bin/bash
# usage: tools/attic/prof_gemm_mem.sh <tag> [prof_gemm.py args]   -> gpurun_out/profmem_<tag>/summary.txt
# memory-path counters of a many-row GEMM launch: VMEM issue, TA / TCP stalls, LDS FIFOs (each --pmc pass is its own run)
set -u
tag=$1; shift
out=$PWD/gpurun_out/profmem_$tag
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/pmc1 -- python3 tools/attic/prof_gemm.py "$@" > $out/pmc1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $out/pmc2 -- python3 tools/attic/prof_gemm.py "$@" > $out/pmc2.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum --output-format csv -d $out/pmc3 -- python3 tools/attic/prof_gemm.py "$@" > $out/pmc3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $out/pmc4 -- python3 tools/attic/prof_gemm.py "$@" > $out/pmc4.log 2>&1 || exit 1
python3 tools/attic/prof_summary.py $out "${FILTER-gemm}" > $out/summary.txt 2>&1
tail -3 $out/pmc*.log >> $out/summary.txt
cat $out/summary.txt
