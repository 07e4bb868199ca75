"""Loader for the CPU oracle (oracle/libngsid_oracle.so).  Test infrastructure - never imported by the product."""
import ctypes, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def load_oracle():
    from ngspeciesid_amd._capi import Api
    path = os.path.join(ROOT, "oracle", "libngsid_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle")) if f.endswith((".c", ".h"))]
    if (not os.path.exists(path)) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return Api(ctypes.CDLL(path), "ongsid_", None)
