timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -5 || exit 1
for f in 1 2 3 4; do echo "== frames in flight $f"; timeout -k 10 120 python tools/quickbench.py --frames-in-flight $f --steps 24 2>&1 | grep "^frame" || exit 1; done
