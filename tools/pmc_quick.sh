#!/bin/bash
# usage: tools/pmc_quick.sh <tag> <counters...>   one --pmc pass over `quickbench.py --steps 5`, per-kernel means printed
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/pmcq_$TAG
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/p -- python3 tools/quickbench.py --steps 5 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 profiles/summarize_pmc.py $OUT | grep -A6 "traceKernel<false, false, false, false>\|traceKernel<true, false, false, false>"
