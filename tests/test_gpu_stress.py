"""Stress parity (GPU): many independent POA / polish groups, HIP vs the CPU oracle, bit for bit.

The small hand-made cases in test_gpu_consensus.py exercise every mode once; rare graph shapes (sibling reuse, far predecessors,
irregular rows, tile splits) only show up in volume.  This is the test that caught the topological-order violation fixed in
oracle g_add_alignment / k_poa phase A ("POA graph order").
"""
import os, subprocess, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "stress_poa.py")
TOOL_ALN = os.path.join(ROOT, "tools", "stress_align.py")


def _run(*args, tool=TOOL, env=None):
    p = subprocess.run([sys.executable, tool] + [str(a) for a in args], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                       env=dict(os.environ, **env) if env else None)
    tail = "\n".join(p.stdout.splitlines()[-8:])
    assert p.returncode == 0, tail


@pytest.mark.gpu
@pytest.mark.parametrize("groups,depth,seed", [(700, 8, 11), (120, 20, 5), (40, 70, 3)])
def test_spoa_groups_match_oracle(groups, depth, seed):
    _run(groups, depth, "spoa", seed)


@pytest.mark.gpu
@pytest.mark.parametrize("groups,depth,seed", [(80, 24, 11), (20, 90, 7)])
def test_polish_groups_match_oracle(groups, depth, seed):
    _run(groups, depth, "polish", seed)


@pytest.mark.gpu
@pytest.mark.parametrize("what,groups,depth,band", [("spoa", 300, 8, 64), ("spoa", 150, 8, 256), ("polish", 40, 24, 64), ("polish", 30, 24, 256)])
def test_other_band_widths_match_oracle(what, groups, depth, band):
    """the 64- and 256-column instances of the tile kernel (one and four band cells per lane)"""
    _run(groups, depth, what, 13, band)


@pytest.mark.gpu
@pytest.mark.parametrize("pairs,maxlen,seed", [(2500, 1100, 1), (300, 4400, 2)])
def test_aligner_pairs_match_oracle(pairs, maxlen, seed):
    """16-bit packed kernel (<= 4000 bases) and the 32-bit kernel, incl. wildcards, lower case, empty and unrelated sequences"""
    _run(pairs, maxlen, seed, tool=TOOL_ALN)


@pytest.mark.gpu
@pytest.mark.parametrize("n_reads,n_species,div,mu,seed", [(5000, 6, 0.15, 17.0, 3), (3000, 12, 0.06, 14.0, 4), (2000, 150, 0.10, 13.0, 5), (2500, 300, 0.03, 15.0, 6), (3200, 1600, 0.30, 15.0, 7)])
def test_cluster_volume_matches_oracle(gpu_api, oracle, n_reads, n_species, div, mu, seed):
    """greedy clustering at a size where speculative blocks, cache hits and database merges all occur; closely related species
    (6 % divergence) force many aligner decisions; the cases with hundreds of species found many representatives per block (several are
    committed per pass: related ones must cut the pass, unrelated ones must not)"""
    import numpy as np
    from ngspeciesid_amd import synth
    from ngspeciesid_amd._capi import ReadSet, cluster_params
    from ngspeciesid_amd.ptable import select_p_table
    sp = synth.make_species(n_species, 700, div, seed=seed)
    rd = synth.make_reads(sp, n_reads, mu=mu, seed=seed + 1, rc_fraction=0.3)
    rs0 = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    score, err, keep = gpu_api.score_reads(rs0, 13, 7.0)
    oscore, oerr, okeep = oracle.score_reads(rs0, 13, 7.0)
    assert np.array_equal(score, oscore) and np.array_equal(keep, okeep)
    from ngspeciesid_amd.hostutil import subset_reads
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    rs = subset_reads(rs0, idx)
    prm = cluster_params(k=13, w=20, p_shared=select_p_table(13, 20))
    ar = np.arange(rs.n, dtype=np.uint32)
    rep, herr, st, cnt = gpu_api.cluster_greedy(rs, prm, acc_rank=ar)
    orep, oherr, ost, ocnt = oracle.cluster_greedy(rs, prm, acc_rank=ar)
    bad = np.nonzero((rep != orep) | (st != ost))[0]
    assert len(bad) == 0, "cluster: %d reads differ, first %s got %s exp %s" % (len(bad), bad[:8], rep[bad[:8]], orep[bad[:8]])
    assert np.array_equal(cnt, ocnt)


TOOL_ED = os.path.join(ROOT, "tools", "stress_ed.py")


@pytest.mark.gpu
@pytest.mark.parametrize("pairs,maxq,seed", [(6000, 1024, 3), (3000, 300, 5), (2500, 770, 4), (900, 2600, 7)])
def test_edit_distance_aligner_matches_oracle(pairs, maxq, seed):
    """bit-parallel polisher aligner (k_ed_align): distance, span and window break points vs the oracle's plain DP; every block-count
    instance (queries up to 256 / 512 / 768 / 1024 and block groups beyond), partial last blocks, wildcards, lower case, empty and unrelated sequences"""
    _run(pairs, maxq, seed, tool=TOOL_ED)


@pytest.mark.gpu
@pytest.mark.parametrize("pairs,maxq,seed,escale", [(1000, 2600, 9, 0.1), (150, 5000, 10, 0.05), (3000, 1300, 12, 0.3), (4000, 760, 14, 0.4)])
def test_edit_distance_full_length_reads_match_oracle(pairs, maxq, seed, escale):
    """the polisher's situation: full-length queries against targets of similar length with a small distance - whole waves stay inside the
    band; queries over 1 024 bases run in the sliding-window instance (register window of 8 blocks, band-relative traceback storage)"""
    _run(pairs, maxq, seed, escale, "full", tool=TOOL_ED)


@pytest.mark.gpu
@pytest.mark.parametrize("pairs,length,err", [(20000, 2000, 0.02), (6000, 5000, 0.05)])
def test_edit_distance_banded_equals_unbanded_at_scale(pairs, length, err):
    """batches large enough for the length-class launches, at sizes the oracle cannot do: the banded / windowed launches (+ the unbanded
    second launch for pairs beyond the band) must return exactly what the unbanded kernel returns"""
    _run(pairs, length, err, tool=os.path.join(ROOT, "tools", "micro", "check_ed_band.py"))


@pytest.mark.gpu
@pytest.mark.parametrize("band", ["12", "40", "300", "0"])
def test_edit_distance_band_does_not_change_results(band):
    """the Ukkonen band of the first launch is exact: with a narrow band most pairs take the unbanded second launch, with a wide one none,
    with 0 the band is off - the oracle comparison must hold for all of them"""
    _run(3000, 900, 8, tool=TOOL_ED, env={"NGSID_OPTIONS": "ed_band=%s" % band})


@pytest.mark.gpu
def test_cluster_result_independent_of_block_size():
    """the speculative-block driver is exact: 120 k reads (several blocks, representatives founded in many of them, noisy singletons)
    must cluster identically with the adaptive block size and with a fixed small one"""
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from ngspeciesid_amd import runtime, synth
from ngspeciesid_amd._capi import ReadSet, cluster_params
from ngspeciesid_amd.ptable import select_p_table
from ngspeciesid_amd.hostutil import subset_reads
api = runtime.get_api(0)
sp = synth.make_species(40, 600, 0.12, seed=21)
ab = np.array([0.5 ** (i / 4.0) for i in range(40)]); ab /= ab.sum()
rd = synth.make_reads(sp, 120000, mu=14.0, seed=22, abundance=ab, rc_fraction=0.2)
rs0 = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
score, err, keep = api.score_reads(rs0, 13, 7.0)
idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
rs = subset_reads(rs0, idx)
prm = cluster_params(k=13, w=20, p_shared=select_p_table(13, 20))
ar = np.arange(rs.n, dtype=np.uint32)
res = []
import ctypes as C
# (block size, cap of a restarting block's rest): adaptive + the default cut, a fixed small block, adaptive with a cut after 500 items (round 6: a block that keeps restarting
# is cut short and its tail decided again by the next block), adaptive without the cut
for blk, trunc in ((0, -1), (3000, -1), (0, 500), (0, 0)):
    assert api.lib.ngsid_ctx_option(api.ctx, b"cluster_block", C.c_int64(blk)) == 0
    assert api.lib.ngsid_ctx_option(api.ctx, b"cluster_trunc", C.c_int64(32768 if trunc < 0 else trunc)) == 0
    rep, herr, st, cnt = api.cluster_greedy(rs, prm, acc_rank=ar)
    res.append((rep.copy(), st.copy(), cnt.copy()))
for x in res[1:]:
    assert np.array_equal(res[0][0], x[0]) and np.array_equal(res[0][1], x[1]) and np.array_equal(res[0][2], x[2]), "block size / block cut changed the result"
print("ok", rs.n, len(np.unique(res[0][0])), res[0][2])
''' % ROOT
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0, "\n".join(p.stdout.splitlines()[-10:])


@pytest.mark.gpu
@pytest.mark.parametrize("trials,seed", [(30, 1), (30, 2)])
def test_whole_pipeline_matches_oracle_on_random_configurations(trials, seed):
    """the WHOLE hot path (score, cluster, draft, rc merge, polish) on random small read sets - species count, length (incl. 507 / 1003: merged
    tail window), depth, error profile, strand mix, k/w (13/20, 15/50, 25/30, 30/35, ...), tile depth, band, iterations, early stop: cluster map,
    counters and every draft / polished sequence identical on the HIP library and the oracle (tools/stress_pipeline.py)"""
    _run(trials, seed, tool=os.path.join(ROOT, "tools", "stress_pipeline.py"))
