"""dev tool: k_hpc_minimizers on 1 M synthetic 750 bp reads resident in HBM (HIP-event time) through the device-pointer C-ABI."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ngspeciesid_amd import runtime, synth
from ngspeciesid_amd._capi import ReadSet
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000; k = int(sys.argv[2]) if len(sys.argv) > 2 else 13; w = int(sys.argv[3]) if len(sys.argv) > 3 else 20
L = int(sys.argv[4]) if len(sys.argv) > 4 else 750
if os.environ.get("NGSID_LIB"): runtime.LIB_PATH = os.environ["NGSID_LIB"]
api = runtime.get_api(0); dev = torch.device("cuda", 0)
sp = synth.make_species(5, L, 0.15, seed=1)
rd = synth.make_reads(sp, n, mu=17.0, seed=2, device=dev)
rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
total = int(rd["off"][-1].item())
moff = torch.zeros(n + 1, dtype=torch.int64, device=dev); codes = torch.zeros(total, dtype=torch.int64, device=dev); pos = torch.zeros(total, dtype=torch.int32, device=dev)
hl = np.zeros(n, dtype=np.uint32); he = np.zeros(n); need = C.c_uint64(0)
torch.cuda.synchronize()
for rep in range(3):
    api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
    rc = api.lib.ngsid_hpc_minimizers(api.ctx, C.byref(rs.c), C.c_int32(k), C.c_int32(w), C.c_void_p(moff.data_ptr()), C.c_void_p(codes.data_ptr()), C.c_void_p(pos.data_ptr()), C.c_uint64(total), C.byref(need),
                                      hl.ctypes.data_as(C.c_void_p), he.ctypes.data_as(C.c_void_p))
    buf = C.create_string_buffer(1 << 12); api.lib.ngsid_profile_read(api.ctx, buf, C.c_uint64(len(buf)))
    ms = [float(l.split()[2]) for l in buf.value.decode().splitlines() if l.startswith("k_hpc_minimizers")][0]
    M = int(need.value)
    print("rc", rc, "k_hpc_minimizers %.3f ms" % ms, "minimizers", M, "algorithmic GB/s %.1f" % ((2.0 * total + 12.0 * M) / ms / 1e6), "hpc_err sum %.9f" % float(he.sum()), flush=True)
