for n in 1000000 100000 10000 1000; do echo "== soup $n"; timeout -k 10 120 python tools/quickbench.py --size $n --steps 10 2>&1 | tail -5 || exit 1; done
