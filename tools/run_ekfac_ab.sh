python tools/prof_ekfac.py 2>/dev/null | tail -3
python -m pytest tests/test_operators_gpu.py tests/test_nets.py tests/test_ggn_diagonal.py -x -q -m gpu -k "ekfac or diagonal or toy" 2>&1 | tail -3
