# usage: tools/sweep_variants.sh name...   ("base" = the default build); prints bench and in-order timings per variant
for v in "$@"; do
  if [ "$v" != "base" ]; then export RT_LIB=$PWD/variants/librt_$v.so; else unset RT_LIB; fi
  timeout -k 10 100 python -u tools/quickbench.py --steps 20 > gpurun_out/sw_q_$v.log 2>&1 || exit 1
  timeout -k 10 100 python -u bench.py --no-cpu --steps 60 > gpurun_out/sw_b_$v.log 2>&1 || exit 1
  q=$(grep "^frame" gpurun_out/sw_q_$v.log | cut -c1-110)
  b=$(python3 -c "import json; d=json.loads([l for l in open('gpurun_out/sw_b_$v.log') if l.startswith('{')][-1]); print(round(d['value'],1))")
  echo "variant '$v': bench $b | $q"
done
