#!/bin/bash
# PMC passes for a command (default: quickbench). Usage: profiles/run_pmc.sh <tag> <cmd...>
set -u
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- $CMD > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
CMD="$*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 profiles/summarize_pmc.py "$OUT"
