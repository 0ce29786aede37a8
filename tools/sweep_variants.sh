# usage: tools/sweep_variants.sh name[:quickbench-args]...   ("base" = the default build); prints bench and in-order timings per variant
for spec in "$@"; do
  v=${spec%%:*}; extra=""; [ "$spec" != "$v" ] && extra=${spec#*:}
  tag=$(echo "$spec" | tr ' :-' '___')
  if [ "$v" != "base" ]; then export RT_LIB=$PWD/variants/librt_$v.so; else unset RT_LIB; fi
  timeout -k 10 100 python -u tools/quickbench.py --steps 20 $extra > gpurun_out/sw_q_$tag.log 2>&1 || { tail -3 gpurun_out/sw_q_$tag.log; exit 1; }
  timeout -k 10 100 python -u tools/quickbench.py --steps 60 --frames-in-flight 12 $extra > gpurun_out/sw_b_$tag.log 2>&1 || { tail -3 gpurun_out/sw_b_$tag.log; exit 1; }
  q=$(grep "^frame" gpurun_out/sw_q_$tag.log | cut -c1-110)
  b=$(grep "^frame" gpurun_out/sw_b_$tag.log | sed 's/.*wall/wall/')
  echo "variant '$spec': in flight: $b | in order: $q"
done
