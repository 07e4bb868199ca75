"""GPU box: tests/test_gpu_configs.py::test_c4_c5_composed_eight_shards_on_one_gpu at a chosen total (round 5: C4 at its full 10 M reads, with the single-process comparison).
    python tools/micro/run_composed_full.py c4 10000000"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import torch
from ngspeciesid_amd import runtime
import test_gpu_configs as T
api = runtime.get_api(0)
name, total = sys.argv[1], int(sys.argv[2])
t = time.time()
T.test_c4_c5_composed_eight_shards_on_one_gpu(api, name, total, compare_single_process=True)
print("OK", name, total, round(time.time() - t, 1), "s; peak torch mem GB", torch.cuda.max_memory_allocated() / 1e9, flush=True)
free, tot = torch.cuda.mem_get_info(); print("free now GB", free / 1e9, "of", tot / 1e9)
