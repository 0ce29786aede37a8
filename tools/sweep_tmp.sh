timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3 || exit 1
for o in 0 1; do for w in 0 16 12; do echo "== tile order $o waves $w"; timeout -k 10 120 python tools/quickbench.py --tile-order $o --waves $w --steps 10 2>&1 | grep "frame\|closest\|shadow:" || exit 1; done; done
