for f in 1 2 4; do echo "== tiny scene, frames in flight $f"; timeout -k 10 120 python tools/quickbench.py --size 1000 --frames-in-flight $f --steps 200 2>&1 | grep "^frame" || exit 1; done
