"""GPU: edge cases of the C-ABI - empty and ragged inputs, short reads, error codes - always against the oracle's behaviour."""
import numpy as np
import pytest
from ngspeciesid_amd._capi import ReadSet, cluster_params, poa_params, polish_params, NgsidError, ST_SHORT, ST_NEWREP
from ngspeciesid_amd.ptable import select_p_table
from ngspeciesid_amd import synth

pytestmark = pytest.mark.gpu
PT = select_p_table(13, 20)


def both(gpu_api, oracle, fn):
    return fn(gpu_api), fn(oracle)


def test_empty_read_set(gpu_api, oracle):
    rs = ReadSet.from_strings([], [])
    for api in (gpu_api, oracle):
        rep, herr, st, cnt = api.cluster_greedy(rs, cluster_params(p_shared=PT))
        assert len(rep) == 0 and cnt.tolist() == [0, 0, 0, 0]
        assert api.poa_consensus(rs, [0], poa_params()) == []
        sc, nc, nm, rg = api.sg_align_batch(rs, rs, [], [], [], 1)
        assert len(sc) == 0


def test_short_and_ragged_reads(gpu_api, oracle):
    seqs = ["A", "ACGT", "ACGTACGTACGTA", "ACGTACGTACGTAC", "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA", "ACGTTGCATGCAAGCTTAGCTAGGCTAGCTAGCATCGATCGATGGCATCGATGCATGCTAGCTAGTCGATCG" * 3,
            "ACGTTGCATGCAAGCTTAGCTAGGCTAGCTAGCATCGATCGATGGCATCGATGCATGCTAGCTAGTCGATCG" * 3, "T" * 40 + "ACGATCGATCGTACGTAGCTAGCTAGCATGCATGCTAGCTAGCTAGCTAGCATGCATCGAT"]
    quals = ["I" * len(s) for s in seqs]
    rs = ReadSet.from_strings(seqs, quals)
    g, o = both(gpu_api, oracle, lambda a: a.cluster_greedy(rs, cluster_params(p_shared=PT)))
    for x, y in zip(g, o):
        assert np.array_equal(x, y, equal_nan=True)
    assert g[2][0] == ST_SHORT and g[2][1] == ST_SHORT and g[2][4] == ST_SHORT       # HPC length < k (cluster.py:266-268)
    assert g[0][6] == 5                                                              # identical read joins the first copy
    gm, om = both(gpu_api, oracle, lambda a: a.hpc_minimizers(rs, 13, 20))
    for x, y in zip(gm, om):
        assert np.array_equal(x, y, equal_nan=True)


def test_error_codes(gpu_api, oracle):
    rs = ReadSet.from_strings(["ACGTRYACGTACGTACGTACGTAGCTAGCTAGCTAGCATCGATCGATCG"], ["I" * 49])
    for api in (gpu_api, oracle):
        with pytest.raises(NgsidError) as e:
            api.cluster_greedy(rs, cluster_params(p_shared=PT))
        assert e.value.code == -3                                                    # NGSID_ERR_ALPHABET
        with pytest.raises(NgsidError) as e:
            api.cluster_greedy(ReadSet.from_strings(["ACGT" * 30], ["I" * 120]), cluster_params(k=33, w=40, p_shared=PT))
        assert e.value.code == -2                                                    # k > 32
    # a (k,w) without rows in the empirical table: KeyError in the reference (cluster.py:367) -> NGSID_ERR_NO_PTABLE
    sp = synth.make_species(1, 300, 0.15, seed=2); rd = synth.make_reads(sp, 20, mu=20.0, seed=3)
    rs2 = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    for api in (gpu_api, oracle):
        with pytest.raises(NgsidError) as e:
            api.cluster_greedy(rs2, cluster_params(p_shared=np.full(225, np.nan)))
        assert e.value.code == -7
    with pytest.raises(NgsidError) as e:
        gpu_api.hpc_minimizers(rs2, 13, 20, cap=3)
    assert e.value.code == -4                                                        # NGSID_ERR_CAPACITY


def test_singletons_and_tiny_groups(gpu_api, oracle):
    sp = synth.make_species(1, 200, 0.15, seed=4); rd = synth.make_reads(sp, 5, mu=20.0, seed=5)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    prm = poa_params(tile_depth=8, band=64)
    goff = [0, 1, 1, 3, 5]                                                           # one read, empty, two reads, two reads
    assert gpu_api.poa_consensus(rs, goff, prm) == oracle.poa_consensus(rs, goff, prm)
    bb = ReadSet.from_strings([rs.get(0)[0], rs.get(3)[0]])
    pp = polish_params(iters=2, tile_depth=8, band=64, trim=2)
    g, o = both(gpu_api, oracle, lambda a: a.polish(bb, rs, [0, 1, 5], pp))           # group 0 has ONE read: < 2 layers -> backbone kept
    assert g[0] == o[0] and np.array_equal(g[1], o[1]) and g[0][0] == rs.get(0)[0]


def test_symmetric_thresholds(gpu_api, oracle):
    g = np.load(__import__("os").path.join(__import__("oracle_lib").GOLD, "cluster_synth600_d10_q14.npz"))
    rs = ReadSet(g["seq"], g["qual"], g["off"])
    prm = cluster_params(k=13, w=20, p_shared=g["p_table"], symmetric=True, mapped_threshold=0.65, aligned_threshold=0.5, min_shared=4, min_fraction=0.7)
    a, b = both(gpu_api, oracle, lambda api: api.cluster_greedy(rs, prm))
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("kw", [dict(min_fraction=0.5), dict(min_fraction=1.0), dict(min_fraction=1.2), dict(min_shared=3, min_fraction=0.9), dict(min_shared=12), dict(mapped_threshold=0.9, aligned_threshold=0.7),
                                dict(symmetric=True, min_fraction=0.6)])
def test_noisy_reads_cluster_as_the_oracle_says_under_other_criteria(gpu_api, oracle, kw):
    """round 6: the speculative driver commits several new representatives per restart round; which later reads a new representative can affect is decided by the walk's own bound
    (nm >= min_shared and not below min(min_fraction, 1) x top, k_cluster.hip rep_can_matter).  A noisy set (many noise representatives of the same species: the case the bound is for)
    under parameter values that move that bound - incl. min_fraction above 1, where only the top count matters - must cluster exactly as the sequential oracle does: map, statuses, counters."""
    sp = synth.make_species(4, 500, 0.12, seed=31)
    rd = synth.make_reads(sp, 6000, mu=12.5, seed=32)
    rs0 = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    from ngspeciesid_amd.hostutil import subset_reads
    score, err, keep = gpu_api.score_reads(rs0, 13, 7.0)
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    rs = subset_reads(rs0, idx)
    prm = cluster_params(k=13, w=20, p_shared=PT, **kw)
    ar = np.arange(rs.n, dtype=np.uint32)
    a, b = both(gpu_api, oracle, lambda api: api.cluster_greedy(rs, prm, acc_rank=ar))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert len(np.unique(a[0])) > 40                    # the set does found many representatives


def test_device_resident_read_set_and_second_context(gpu_api, oracle):
    """ngsid_reads_upload: one copy to HBM serves several calls and several contexts; results equal the host read set's; a second context
    (own stream and scratch) working from another host thread at the same time gives the same answers."""
    import threading
    from ngspeciesid_amd import runtime
    sp = synth.make_species(3, 500, 0.15, seed=5)
    rd = synth.make_reads(sp, 3000, mu=17.0, seed=9)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    prm = cluster_params(k=13, w=20, p_shared=PT)
    dev = gpu_api.upload_reads(rs)
    assert dev is not rs and dev.n == rs.n
    a = gpu_api.cluster_greedy(rs, prm); b = gpu_api.cluster_greedy(dev, prm)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    groups = [np.nonzero(a[0] == r)[0].astype(np.uint32) for r in np.unique(a[0]) if (a[0] == r).sum() >= 50]
    off = np.concatenate(([0], np.cumsum([len(g) for g in groups]))).astype(np.uint64); ro = np.concatenate(groups)
    pp = poa_params(mode=0, match=5, mismatch=-4, gap=-2, tile_depth=8, band=64, trim=1)
    ref = gpu_api.poa_consensus(rs, off, pp, read_order=ro)
    api2 = runtime.new_api(0)
    out = [None, None]
    def work(k, api):
        out[k] = [api.poa_consensus(dev, off, pp, read_order=ro) for _ in range(3)]
    th = [threading.Thread(target=work, args=(0, gpu_api)), threading.Thread(target=work, args=(1, api2))]
    [t.start() for t in th]; [t.join() for t in th]
    assert all(x == ref for x in out[0]) and all(x == ref for x in out[1])
    api2.close(); api2.close()                         # Api.close() = ngsid_destroy, idempotent
    dev.release(); dev.release()                       # idempotent
    with pytest.raises(NgsidError):
        gpu_api._err(gpu_api._call("reads_release", __import__("ctypes").byref(rs.c)))          # a host read set is refused
