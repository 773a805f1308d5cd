for s in 1 2 3; do CLO_FUZZ_WIDE=0.4 timeout 600 python tools/fuzz_native.py $s 120 2>&1 | tail -8; done
