# soak run of every randomised comparison (seeds from $1, default 100)
S=${1:-100}
for k in 0 1 2; do s=$((S + k))
  timeout 600 python tools/fuzz_eigh.py $s 150 2>&1 | tail -3 | sed "s/^/eigh($s): /"
  CLO_FUZZ_EIGH_DEFAULT=1 timeout 600 python tools/fuzz_eigh.py $s 150 2>&1 | tail -3 | sed "s/^/eigh-default($s): /"
  timeout 900 python tools/fuzz_kfac.py $s 60 2>&1 | tail -3 | sed "s/^/kfac($s): /"
  CLO_FUZZ_BIG=1 timeout 900 python tools/fuzz_kfac.py $s 30 2>&1 | tail -3 | sed "s/^/kfac-big($s): /"
  timeout 900 python tools/fuzz_ops.py $s 50 2>&1 | tail -3 | sed "s/^/ops($s): /"
  CLO_FUZZ_WIDE=0.3 timeout 600 python tools/fuzz_native.py $s 150 2>&1 | tail -2 | sed "s/^/native($s): /"
  timeout 600 python tools/fuzz_cholesky.py $s 100 2>&1 | tail -2 | sed "s/^/cholesky($s): /"
  timeout 600 python tools/fuzz_conv_factors.py $s 150 2>&1 | tail -2 | sed "s/^/conv($s): /"
  timeout 600 python tools/fuzz_gemm.py $s 150 2>&1 | tail -2 | sed "s/^/gemm($s): /"
done
