"""bench.py with a different number of eigensolver workers: python tools/bench_workers.py W [bench args]"""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
w = int(sys.argv[1]); sys.argv = ["bench.py"] + sys.argv[2:]
from curvlinops_amd import linalg_native
linalg_native.EIGH_WORKERS = w
if os.environ.get('INV_WORKERS'):
    linalg_native.INVERSE_WORKERS = int(os.environ['INV_WORKERS'])
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
