for v in n1 n8 n16 n24 n32; do echo "== $v"; RT_LIB=variants/librt_$v.so timeout -k 10 120 python tools/quickbench.py --steps 10 2>&1 | grep "^frame\|closest: wave\|shadow: wave" || exit 1; done
