"""The two-pairs-per-wave instance of the int16 aligner (csrc/k_align16p.hip; queries of up to 896 bases in batches of >= 4 096 pairs).
It must return exactly what the one-pair kernel (ngsid_ctx_option align_paired = 0) and the oracle return: score, alignment columns, matches and the
k-window region count depend on the traceback tie-breaks, so equality of all four pins the whole alignment.
"""
import ctypes as C
import numpy as np
import pytest
from ngspeciesid_amd._capi import ReadSet

pytestmark = pytest.mark.gpu
LET = np.frombuffer(b"ACGT", dtype=np.uint8)


def _mutate(rng, s, rate):
    out = []
    for c in s:
        u = rng.random()
        if u < rate * 0.4: out.append(int(LET[rng.integers(0, 4)]))                     # substitution
        elif u < rate * 0.7: continue                                                   # deletion
        elif u < rate: out.append(int(c)); out.append(int(LET[rng.integers(0, 4)]))     # insertion
        else: out.append(int(c))
    return np.array(out, dtype=np.uint8)


def _make(seed, npairs, nt=24, lo_len=500, min_q=513):
    rng = np.random.default_rng(seed)
    targets = [LET[rng.integers(0, 4, int(rng.integers(lo_len, 930)))] for _ in range(nt)]
    targets.append(np.zeros(0, dtype=np.uint8))                                        # an empty target: the degenerate branch of the kernel
    qs, qi, ti = [], [], []
    for p in range(npairs):
        t = int(rng.integers(0, nt))
        base = targets[t] if rng.random() < 0.8 else targets[int(rng.integers(0, nt))]  # 20 % unrelated pairs
        q = _mutate(rng, base, float(rng.choice([0.02, 0.1, 0.2])))
        lo = int(rng.integers(0, 40)); hi = len(q) - int(rng.integers(0, 40))
        q = q[lo:max(hi, lo + min_q + 7)][:896]
        if len(q) < min_q: q = np.concatenate([q, LET[rng.integers(0, 4, min_q - len(q))]])
        if rng.random() < 0.05: q = q.copy(); q[rng.integers(0, len(q), 3)] = ord("N")   # wildcards
        if rng.random() < 0.05: q = q.copy(); q[: len(q) // 3] |= 0x20                    # lower case (raw-character identity differs from the score's)
        if min_q <= 1 and rng.random() < 0.003: q = np.zeros(0, dtype=np.uint8)          # an empty QUERY (class 0): the degenerate branch, query side
        qs.append(q); qi.append(p); ti.append(t if rng.random() > 0.002 else nt)
    Q = ReadSet(np.concatenate(qs), None, np.concatenate(([0], np.cumsum([len(x) for x in qs]))).astype(np.uint64))
    T = ReadSet(np.concatenate(targets), None, np.concatenate(([0], np.cumsum([len(x) for x in targets]))).astype(np.uint64))
    opens = rng.integers(2, 6, npairs).astype(np.int32)
    return Q, T, np.array(qi, dtype=np.uint32), np.array(ti, dtype=np.uint32), opens


def _opt(api, name, v):
    assert api.lib.ngsid_ctx_option(api.ctx, name, C.c_int64(v)) == 0


@pytest.mark.parametrize("seed,lo_len,min_q", [(3, 500, 513), (4, 500, 513), (5, 40, 1), (6, 200, 180)])
def test_paired_kernel_equals_one_pair_kernel_and_oracle(gpu_api, oracle, seed, lo_len, min_q):
    """(3, 4): the 513 - 896 classes; (5, 6): all four single-strip classes, incl. queries of a few bases"""
    Q, T, qi, ti, opens = _make(seed, 6000, lo_len=lo_len, min_q=min_q)
    try:
        _opt(gpu_api, b"align_paired", 1)
        a = gpu_api.sg_align_batch(Q, T, qi, ti, opens, k=13)
        _opt(gpu_api, b"align_paired", 0)
        b = gpu_api.sg_align_batch(Q, T, qi, ti, opens, k=13)
    finally:
        _opt(gpu_api, b"align_paired", 1)
    names = ("score", "ncols", "nmatch", "region")
    for x, y, nm in zip(a, b, names):
        bad = np.nonzero(x != y)[0]
        assert len(bad) == 0, "%s differs between the paired and the one-pair kernel at pairs %s (seed %d)" % (nm, bad[:8].tolist(), seed)
    sub = np.random.default_rng(seed).choice(len(qi), 300, replace=False)
    o = oracle.sg_align_batch(Q, T, qi[sub], ti[sub], opens[sub], k=13)
    for x, y, nm in zip(a, o, names):
        assert np.array_equal(x[sub], y), "%s differs from the oracle (seed %d)" % (nm, seed)


def test_paired_kernel_odd_bins_and_single_class(gpu_api):
    """all queries of one length (one residue bin, odd count: the last item holds one pair) and a batch that only has the 769 - 896 class"""
    rng = np.random.default_rng(9)
    t = LET[rng.integers(0, 4, 800)]
    for qlen, npairs in ((700, 4097), (850, 4099)):
        qs = [_mutate(rng, t, 0.1)[:qlen] for _ in range(npairs)]
        qs = [np.concatenate([q, LET[rng.integers(0, 4, qlen - len(q))]]) if len(q) < qlen else q for q in qs]
        Q = ReadSet(np.concatenate(qs), None, (np.arange(npairs + 1) * qlen).astype(np.uint64)); T = ReadSet(t, None, np.array([0, len(t)], dtype=np.uint64))
        qi = np.arange(npairs, dtype=np.uint32); ti = np.zeros(npairs, dtype=np.uint32)
        try:
            _opt(gpu_api, b"align_paired", 1); a = gpu_api.sg_align_batch(Q, T, qi, ti, 3, k=13)
            _opt(gpu_api, b"align_paired", 0); b = gpu_api.sg_align_batch(Q, T, qi, ti, 3, k=13)
        finally:
            _opt(gpu_api, b"align_paired", 1)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_paired_kernel_break_points_and_spans(gpu_api):
    """the per-window break points and the aligned span (AlignJob.bp / .span: what the polisher's layers are cut from) of the paired kernel against the
    one-pair kernel: a polishing call with the affine read -> backbone aligner (aln_mode 0) over > 4 096 reads returns the same sequences and read
    counts either way (the two outputs were not compared directly)"""
    from ngspeciesid_amd import synth
    from ngspeciesid_amd._capi import polish_params
    sp = synth.make_species(2, 700, 0.15, seed=5)
    rd = synth.make_reads(sp, 5000, mu=15.0, seed=6)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    spc = rd["species"].numpy(); order = np.argsort(spc, kind="stable").astype(np.uint32); n0 = int((spc == 0).sum())
    bb = ReadSet.from_strings([rs.get(int(order[0]))[0], rs.get(int(order[n0]))[0]])
    res = {}
    try:
        for v in (1, 0):
            _opt(gpu_api, b"align_paired", v)
            res[v] = gpu_api.polish(bb, rs, [0, n0, rs.n], polish_params(iters=2, k=13, w=20, tile_depth=6, band=0, trim=2, aln_mode=0, stop_when_stable=0), read_order=order)
    finally:
        _opt(gpu_api, b"align_paired", 1)
    assert res[1][0] == res[0][0] and np.array_equal(res[1][1], res[0][1])
    assert sorted(res[1][0]) == sorted(s.tobytes().decode() for s in sp)



def test_paired_kernel_extreme_scores_and_many_wildcards(gpu_api, oracle):
    """the traceback flags of the paired kernel are sign bits of packed 16-bit differences - exact while nothing wraps (static_assert in k_align16p.hip).  The
    corner of ngsid_align16_applicable: match 4, mismatch -8, open 16, ext 4, identical 896-base queries (score 3 584), unrelated pairs (long gap runs from the floor),
    and reads that are a third wildcards (N scores 0 against everything, also against N)."""
    rng = np.random.default_rng(31)
    npairs = 4200
    targets = [LET[rng.integers(0, 4, 896)] for _ in range(12)]
    qs, ti = [], []
    for p in range(npairs):
        t = int(rng.integers(0, 12)); u = rng.random()
        if u < 0.25: q = targets[t].copy()                                             # identical: the largest score
        elif u < 0.5: q = LET[rng.integers(0, 4, 896)]                                  # unrelated
        else: q = _mutate(rng, targets[t], 0.15)[:896]
        if len(q) < 520: q = np.concatenate([q, LET[rng.integers(0, 4, 520 - len(q))]])
        if rng.random() < 0.5: q = q.copy(); q[rng.random(len(q)) < 0.33] = ord("N")
        qs.append(q); ti.append(t)
    tw = [t.copy() for t in targets]
    for t in tw[:4]: t[rng.random(len(t)) < 0.33] = ord("N")
    Q = ReadSet(np.concatenate(qs), None, np.concatenate(([0], np.cumsum([len(x) for x in qs]))).astype(np.uint64))
    T = ReadSet(np.concatenate(tw), None, np.concatenate(([0], np.cumsum([len(x) for x in tw]))).astype(np.uint64))
    qi = np.arange(npairs, dtype=np.uint32); ti = np.array(ti, dtype=np.uint32)
    kw = dict(ext=4, match=4, mismatch=-8, k=13)
    try:
        _opt(gpu_api, b"align_paired", 1); a = gpu_api.sg_align_batch(Q, T, qi, ti, 16, **kw)
        _opt(gpu_api, b"align_paired", 0); b = gpu_api.sg_align_batch(Q, T, qi, ti, 16, **kw)
    finally:
        _opt(gpu_api, b"align_paired", 1)
    for x, y, nm in zip(a, b, ("score", "ncols", "nmatch", "region")):
        assert np.array_equal(x, y), nm
    assert a[0].max() == 4 * 896
    sub = rng.choice(npairs, 200, replace=False)
    o = oracle.sg_align_batch(Q, T, qi[sub], ti[sub], 16, **kw)
    for x, y, nm in zip(a, o, ("score", "ncols", "nmatch", "region")):
        assert np.array_equal(x[sub], y), nm
