export RT_TRACE_DEBUG=1
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3 || exit 1
for l in 4 8 16; do echo "== default lds $l"; timeout -k 10 120 python tools/quickbench.py --lds-stack $l --steps 10 2>&1 | grep "frame\|<0,0>\|<1,0>\|closest\|shadow:" || exit 1; done
for v in p16t4 p16t12 p32t8 p32t16; do for l in 8; do echo "== $v lds $l"; RT_LIB=variants/librt_$v.so timeout -k 10 120 python tools/quickbench.py --lds-stack $l --steps 10 2>&1 | grep "frame\|<0,0>\|<1,0>" || exit 1; done; done
