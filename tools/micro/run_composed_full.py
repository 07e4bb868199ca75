import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.chdir('/root/repo')
import torch
from ngspeciesid_amd import runtime
import test_gpu_configs as T
api = runtime.get_api(0)
name, total = sys.argv[1], int(sys.argv[2])
import time; t=time.time()
T.test_c4_c5_composed_eight_shards_on_one_gpu(api, name, total, compare_single_process=(total <= 4000000), out_slots=(2 if total > 4000000 else 0))
print("OK", name, total, round(time.time()-t,1), "s; peak torch mem GB", torch.cuda.max_memory_allocated()/1e9, flush=True)
free, tot = torch.cuda.mem_get_info(); print("free now GB", free/1e9, "of", tot/1e9)
