"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/ngsid.h declares."""
import os, re, ctypes
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ngsid.h")).read()
    return sorted(set(re.findall(r"\b(ngsid_[a-z_0-9]+)\s*\(", txt)))


def test_exports_all_declared_symbols():
    from ngspeciesid_amd import runtime
    if not os.path.exists(runtime.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = runtime.load_library()
    names = _declared()
    assert "ngsid_cluster_greedy" in names and "ngsid_polish" in names
    for n in names:
        assert hasattr(lib, n), "libngsid_hip.so does not export %s" % n
    assert lib.ngsid_abi_version() == 2


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ngspeciesid_amd import runtime
    from ngspeciesid_amd._capi import NgsidError
    with pytest.raises(NgsidError):
        runtime.get_api(0)


def test_oracle_twins_exist(oracle):
    for n in _declared():
        if n in ("ngsid_create", "ngsid_destroy", "ngsid_profile_enable", "ngsid_profile_read", "ngsid_ctx_option", "ngsid_reads_upload", "ngsid_reads_release", "ngsid_reads_subset") or n.startswith("ngsid_host_"):      # device management / host-only helpers: tested directly (test_fastio_cpu.py, test_gpu_edge.py)
            continue
        assert hasattr(oracle.lib, "o" + n), "oracle lacks o%s" % n
