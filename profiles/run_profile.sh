#!/bin/bash
# Collects the rocprofv3 evidence for the bench kernel on one MI355X:
#   1. kernel trace + stats (per-kernel average duration)
#   2..n  one --pmc pass each (SQ issue/wait mix, L2 hit/miss, HBM fetch, HBM write)
# Usage: profiles/run_profile.sh <tag> [bench args...]   (run from the repo root on the GPU box)
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# the profiler initialises HIP before bench.py runs, so the queue count bench.py asks for is set here
export GPU_MAX_HW_QUEUES=32
# one frame in flight: the kernels run alone, as in the part of bench.py that measures the roofline
ARGS="--steps 16 --warmup 4 --no-cpu --no-others --frames-in-flight 1 $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass ta TA_TA_BUSY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum
# one run writes the summary and the table bench.py reads (copy both, and trace/*_kernel_stats.csv, into profiles/<round>/)
WORKLOAD=${WORKLOAD:-soup-1000000-1920x1080-l1-d0}
python3 profiles/summarize_pmc.py "$OUT" --traffic-json "$OUT/traffic.json" --workload "$WORKLOAD" --summary-out "$OUT/pmc_summary.txt" > /dev/null 2>&1
tail -60 "$OUT/pmc_summary.txt"
echo done
