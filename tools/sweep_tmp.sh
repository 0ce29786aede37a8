export GPU_MAX_HW_QUEUES=16
for v in k6 k8; do lib=variants/librt_$v.so; echo "== $v"; for a in "--stride 8 --frames-in-flight 8" "--stride 4 --frames-in-flight 8" "--stride 2 --frames-in-flight 8" "--stride 1 --frames-in-flight 4" "--stride 1 --frames-in-flight 1"; do echo "$a"; RT_LIB=$lib timeout -k 10 120 python tools/quickbench.py --steps 48 $a 2>&1 | grep "^frame" || exit 1; done; done
