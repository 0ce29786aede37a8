for v in popif; do echo "== $v"; RT_LIB=variants/librt_$v.so timeout -k 10 120 python tools/quickbench.py --steps 10 2>&1 | grep "frame\|closest\|shadow:" || exit 1; done
echo "== default"; timeout -k 10 120 python tools/quickbench.py --steps 10 2>&1 | grep "frame\|closest\|shadow:"
