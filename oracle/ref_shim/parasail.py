"""parasail-shaped shim (TEST INFRASTRUCTURE, build container only).

The reference imports `parasail` (cluster.py:11, consensus.py:7); parasail 1.2.4 is not installed and
cannot be installed here.  This module exposes the three names the reference uses
(matrix_create, sg_trace_scan_16, sg_trace_scan_32) backed by the oracle's own semi-global aligner
(oracle/ngsid_oracle.c: sg_align), so that the reference's Python can be executed to pin the rest of
the oracle.  Tie-breaking of the traceback is therefore the oracle's, not parasail's (parity unpinned).
"""
import ctypes, os
_lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "libngsid_oracle.so"))
_lib.ongsid_sg_align_cigar.restype = ctypes.c_int32
CALLS = []          # (len(s1), len(s2), open, ext) log used by make_golden.py


class _Cigar:
    def __init__(self, text): self.decode = text


class _Result:
    def __init__(self, score, cigar):
        self.saturated = False
        self.score = score
        self.cigar = _Cigar(cigar)


def matrix_create(alphabet, match, mismatch):
    assert alphabet == "ACGT"
    return (match, mismatch)


def sg_trace_scan_16(s1, s2, open_, ext, matrix):
    b1 = s1.encode(); b2 = s2.encode()
    cap = 16 * (len(b1) + len(b2)) + 64
    buf = ctypes.create_string_buffer(cap)
    score = ctypes.c_int32(0)
    rc = _lib.ongsid_sg_align_cigar(b1, len(b1), b2, len(b2), matrix[0], matrix[1], int(open_), int(ext), buf, cap, ctypes.byref(score))
    assert rc == 0
    CALLS.append((len(b1), len(b2), int(open_), int(ext)))
    return _Result(score.value, buf.value)


sg_trace_scan_32 = sg_trace_scan_16
