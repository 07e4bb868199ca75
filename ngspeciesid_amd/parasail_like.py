"""A `parasail`-shaped front of the HIP aligner for callers written against the reference's use of parasail
(cluster.py:130-144: `parasail.sg_trace_scan_16/32(s1, s2, open, extend, matrix)` -> result.saturated / result.score / result.cigar.decode;
consensus.py:58-63).  Same scoring conventions: semi-global with all end gaps free, a gap of length L costs open + (L-1) * extend, the
substitution matrix is (match, mismatch) over ACGT.  One call = one pair on the GPU (use Api.sg_align_cigar_batch for batches).
"""
from __future__ import annotations
from . import runtime
from ._capi import ReadSet


def matrix_create(alphabet, match, mismatch):
    return (int(match), int(mismatch))


class _Cigar:
    def __init__(self, ops):
        out, i = [], 0
        while i < len(ops):
            j = i
            while j < len(ops) and ops[j] == ops[i]:
                j += 1
            out.append("%d%s" % (j - i, ops[i])); i = j
        self.decode = "".join(out).encode()
        self.ops = ops


class Result:
    def __init__(self, score, ops):
        self.score, self.saturated, self.cigar = int(score), False, _Cigar(ops)        # scores are computed in 32 bits: never saturated


def _align(s1, s2, open_, extend, matrix, api=None):
    api = api or runtime.get_api()
    score, ops = api.sg_align_cigar_batch(ReadSet.from_strings([s1]), ReadSet.from_strings([s2]), [0], [0], int(open_), ext=int(extend), match=matrix[0], mismatch=matrix[1])
    return Result(score[0], ops[0])


def sg_trace_scan_16(s1, s2, open_, extend, matrix, api=None):
    return _align(s1, s2, open_, extend, matrix, api)


sg_trace_scan_32 = sg_trace_scan_16
sg_trace_scan_sat = sg_trace_scan_16
