"""Loads libngsid_hip.so (the MI355X hot path) and owns the per-process context.

There is deliberately no CPU fallback: if the shared library or a HIP device is missing this raises.
"""
from __future__ import annotations
import ctypes as C
import os
from ._capi import Api, NgsidError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libngsid_hip.so")
_lib = None
_apis = {}


def load_library() -> C.CDLL:
    """dlopen the C-ABI library (works without a GPU: used by the CPU test that checks exported symbols)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libngsid_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the hot path.")
        _lib = C.CDLL(LIB_PATH)
        _lib.ngsid_last_error.restype = C.c_char_p
        _lib.ngsid_abi_version.restype = C.c_uint32
    return _lib


def get_api(device: int | None = None) -> Api:
    """Context bound to `device` (default: LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if device not in _apis:
        lib = load_library()
        ctx = C.c_void_p()
        rc = lib.ngsid_create(C.c_int32(device), C.c_uint32(0), C.byref(ctx))
        if rc != 0:
            raise NgsidError(rc, (lib.ngsid_last_error(None) or b"").decode())
        _apis[device] = Api(lib, "ngsid_", ctx)
        _apis[device].device = device
        _apis[device]._options = {}
        # test / tool hook: NGSID_OPTIONS="ed_band=12,cluster_block=3000" -> ngsid_ctx_option calls on the new context (the LIBRARY reads no environment)
        for kv in filter(None, os.environ.get("NGSID_OPTIONS", "").split(",")):
            name, val = kv.split("=")
            _apis[device]._options[name.strip()] = int(val)
            rc = lib.ngsid_ctx_option(ctx, name.strip().encode(), C.c_int64(int(val)))
            if rc != 0:
                raise NgsidError(rc, (lib.ngsid_last_error(ctx) or b"").decode())
    return _apis[device]


def new_api(device: int | None = None, options: dict | None = None) -> Api:
    """A fresh, uncached context on `device` (own HIP stream, own scratch buffers): concurrent pipelines of one process use one each."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    lib = load_library()
    ctx = C.c_void_p()
    rc = lib.ngsid_create(C.c_int32(device), C.c_uint32(0), C.byref(ctx))
    if rc != 0:
        raise NgsidError(rc, (lib.ngsid_last_error(None) or b"").decode())
    if options is None:          # (the same test / tool hook as get_api)
        options = {kv.split("=")[0].strip(): int(kv.split("=")[1]) for kv in filter(None, os.environ.get("NGSID_OPTIONS", "").split(","))}
    for name, val in (options or {}).items():
        rc = lib.ngsid_ctx_option(ctx, name.encode(), C.c_int64(int(val)))
        if rc != 0:
            raise NgsidError(rc, (lib.ngsid_last_error(ctx) or b"").decode())
    api = Api(lib, "ngsid_", ctx)
    api.device = device
    api.lanes = 1                # a caller that makes contexts of its own runs its own concurrent pipelines (virtual ranks, tools): no lanes below them (_capi.Api.lanes)
    api._options = dict(options or {})
    return api
