"""CPU: the dealing and the merging of a call that is split over lanes (_capi.Api lanes), with the oracle standing in for both contexts."""
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from ngspeciesid_amd import _capi
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, POA_LOCAL, lane_deal
from test_consensus_oracle import make_set


def test_deal():
    assert lane_deal([5, 5, 5, 5, 5], 2) == [[0, 2, 4], [1, 3]]
    assert lane_deal([1, 9, 1, 1], 2) == [[1], [0, 2, 3]]
    assert lane_deal([4], 2) == [[0]]
    assert sorted(sum(lane_deal([3, 1, 4, 1, 5, 9, 2, 6], 3), [])) == list(range(8))
    off, ro = _capi._lane_lists(np.array([0, 2, 5, 9]), None, [0, 2])
    assert list(off) == [0, 2, 6] and list(ro) == [0, 1, 5, 6, 7, 8]
    off, ro = _capi._lane_lists(np.array([0, 2, 5, 9]), np.arange(9)[::-1], [1])
    assert list(off) == [0, 3] and list(ro) == [6, 5, 4]
    bb = _capi._lane_backbones(ReadSet.from_strings(["AAA", "CC", "GGGG"]), [2, 0])
    assert bb.seq.tobytes() == b"GGGGAAA" and list(bb.off) == [0, 4, 7]


def test_split_calls_merge_to_the_unsplit_result(oracle, monkeypatch):
    sp, rd, rs = make_set(240, L=420, mu=14.0, seed=9, nsp=3)
    spc = rd["species"].numpy(); order = np.argsort(spc, kind="stable").astype(np.uint32)
    cnt = np.bincount(spc); off = np.concatenate(([0], np.cumsum(cnt))).astype(np.uint64)
    bb = ReadSet.from_strings([rs.get(int(order[int(off[g])]))[0] for g in range(3)])
    prm = polish_params(iters=2, k=13, w=20, tile_depth=4, band=0, trim=2)
    pp = poa_params(mode=POA_LOCAL, tile_depth=4, band=0, trim=1)
    want = (oracle.polish(bb, rs, off, prm, read_order=order), oracle.polish_trace(bb, rs, off, prm, read_order=order, aln=True), oracle.poa_consensus(rs, off, pp, read_order=order))
    ex = ThreadPoolExecutor(max_workers=1)
    monkeypatch.setattr(_capi.Api, "_lane_deal", lambda self, rs_, go, backbones=None: lane_deal(np.diff(np.asarray(go, dtype=np.int64)), 2))
    monkeypatch.setattr(_capi.Api, "_twins", lambda self, n: [(self, ex)] * n)
    got = (oracle.polish(bb, rs, off, prm, read_order=order), oracle.polish_trace(bb, rs, off, prm, read_order=order, aln=True), oracle.poa_consensus(rs, off, pp, read_order=order))
    ex.shutdown()
    assert got[0][0] == want[0][0] and np.array_equal(got[0][1], want[0][1])
    assert got[1][0] == want[1][0] and np.array_equal(got[1][1], want[1][1]) and np.array_equal(got[1][2], want[1][2])
    assert got[2] == want[2]
    rec = got[1][2]; go = off.astype(np.int64)                       # the records of a lane call: a group's range is a view into the lane's array, anything else the merged one
    assert isinstance(rec, _capi.LaneRecords) and rec.shape == want[1][2].shape and len(rec) == 2
    for it in range(2):
        for g in range(3): assert np.array_equal(rec[it][int(go[g]):int(go[g + 1])], want[1][2][it, int(go[g]):int(go[g + 1])])
        assert np.array_equal(rec[it][3:17], want[1][2][it, 3:17]) and np.array_equal(np.asarray(rec[it]), want[1][2][it])


def test_a_lane_out_of_memory_falls_back_to_one_context(oracle, monkeypatch):
    """a second context's working set that does not fit beside the first (NGSID_ERR_HIP "out of memory" from a lane): the call is repeated in one context, lanes stay off"""
    from ngspeciesid_amd._capi import NgsidError
    sp, rd, rs = make_set(120, L=300, mu=15.0, seed=2, nsp=2)
    spc = rd["species"].numpy(); order = np.argsort(spc, kind="stable").astype(np.uint32)
    off = np.concatenate(([0], np.cumsum(np.bincount(spc)))).astype(np.uint64)
    bb = ReadSet.from_strings([rs.get(int(order[int(off[g])]))[0] for g in range(2)])
    prm = polish_params(iters=1, k=13, w=20, tile_depth=4, band=0, trim=2)
    want = oracle.polish(bb, rs, off, prm, read_order=order)
    def boom(self, deal, fn): raise NgsidError(-6, "poa_host.hip:1 hipMalloc -> out of memory")
    monkeypatch.setattr(_capi.Api, "_lane_deal", lambda self, rs_, go, backbones=None: lane_deal(np.diff(np.asarray(go, dtype=np.int64)), 2))
    monkeypatch.setattr(_capi.Api, "_lane_run", boom)
    import pytest
    try:
        oracle.lanes = 2
        got = oracle.polish(bb, rs, off, prm, read_order=order)
        assert got[0] == want[0] and oracle.lanes == 1
        def other(self, deal, fn): raise NgsidError(-3, "base outside ACGTN")
        monkeypatch.setattr(_capi.Api, "_lane_run", other); oracle.lanes = 2
        with pytest.raises(NgsidError): oracle.polish(bb, rs, off, prm, read_order=order)
    finally:
        oracle.__dict__.pop("lanes", None)
