"""GPU, BASELINE.json's full single-GPU sizes: size-independent properties of the whole hot path (the oracle cannot run these sizes).

C3 = 1 M x 750 bp ONT-profile reads, 5 species (the bench workload); C2 = 100 k x 750 bp, 1 species; C5-like = 200 k x 2 kb CCS,
20 species, k15/w50.  Properties: clusters are pure and complete, the representative map is idempotent and points backwards in
processing order, every consensus equals its generating amplicon (tolerance of the north star: <= 1 edit per 10 kb), polishing a
polished sequence again returns it unchanged (fixed point), and the early stop returns what all iterations return.
"""
import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _run(gpu_api, n, nsp, L, mu, k, w, ab, seed, stop=False):
    import torch
    import bench
    from ngspeciesid_amd import pipeline
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.ptable import select_p_table
    dev = torch.device("cuda", 0)
    sp, rd = bench.gen_sorted_reads(gpu_api, n, nsp, L, mu, seed=seed, device=dev)
    rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
    res = pipeline.run_hot_path(gpu_api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=k, w=w, abundance_ratio=ab, racon_iter=3,
                                tile_depth=pipeline.TILE_DEPTH, band=0, p_shared=select_p_table(k, w), polish_stop_when_stable=stop)      # the depth bench.py / the CLI ship with
    return sp, rd, rs, res


def _check_clusters(rd, res, nsp, min_big_fraction):
    spc = rd["species"].cpu().numpy(); rep = res["rep_of"]; n = len(rep)
    assert np.array_equal(rep[rep], rep), "representative map is not idempotent"
    assert np.all(rep <= np.arange(n)), "a read joined a representative that comes later in the processing order"
    reps, counts = np.unique(rep, return_counts=True)
    big = reps[np.argsort(-counts)[:nsp]]
    inbig = np.isin(rep, big)
    assert inbig.mean() >= min_big_fraction, "only %.4f of the reads are in the %d largest clusters" % (inbig.mean(), nsp)
    assert np.array_equal(spc[rep[inbig]], spc[inbig]), "a large cluster mixes species"
    assert len(np.unique(spc[big])) == nsp, "two of the large clusters are the same species"


def _check_consensus(sp, res, max_ed):
    from util_seq import edit_distance
    truths = [s.tobytes().decode() for s in sp]
    assert len(res["centers"]) == len(truths)
    tset = set(truths)
    for c in res["centers"]:
        if c[3] in tset: continue                              # exact hit (the rule at these sizes; 20 truths x 9 end trims of a 2 kb edit distance take a minute)
        ed = min(min(edit_distance(c[3][a:len(c[3]) - b if b else None], t) for a in range(3) for b in range(3)) for t in truths)
        assert ed <= max_ed, "consensus of cluster %d is %d edits away from every amplicon" % (c[1], ed)


def test_c3_one_million_reads_five_species(gpu_api):
    from ngspeciesid_amd import pipeline
    from ngspeciesid_amd._capi import ReadSet, polish_params
    sp, rd, rs, res = _run(gpu_api, 1000000, 5, 750, 17.0, 13, 20, 0.02, seed=7)
    _check_clusters(rd, res, 5, 0.995)
    _check_consensus(sp, res, 0)
    # fixed point: polishing the polished sequences once more (same reads per cluster) changes nothing, and n_used is the cluster's read count +- filters
    rep = res["rep_of"]
    order = np.argsort(rep, kind="stable").astype(np.uint32)
    srt = rep[order]
    p_order, p_off = [], [0]
    for c in res["centers"]:
        ids = np.concatenate([order[np.searchsorted(srt, g, "left"):np.searchsorted(srt, g, "right")] for g in c[4]])
        p_order.append(ids); p_off.append(p_off[-1] + len(ids))
    bb = ReadSet.from_strings([c[3] for c in res["centers"]])
    again, used = gpu_api.polish(bb, rs, p_off, polish_params(iters=1, k=13, w=20, tile_depth=pipeline.TILE_DEPTH, band=128, trim=2), read_order=np.concatenate(p_order))
    assert again == [c[3] for c in res["centers"]], "the polished sequences are not a fixed point of the polisher"
    assert np.all(used >= 0.95 * np.diff(p_off))
    # the early stop is exact at this size too
    sp2, rd2, rs2, res2 = _run(gpu_api, 1000000, 5, 750, 17.0, 13, 20, 0.02, seed=7, stop=True)
    assert np.array_equal(res2["rep_of"], res["rep_of"]) and [c[3] for c in res2["centers"]] == [c[3] for c in res["centers"]]


def test_c2_hundred_thousand_reads_one_species(gpu_api):
    sp, rd, rs, res = _run(gpu_api, 100000, 1, 750, 17.0, 13, 20, 0.1, seed=11)
    _check_clusters(rd, res, 1, 0.995)
    _check_consensus(sp, res, 0)


def test_c5_like_ccs_two_kb_twenty_species(gpu_api):
    sp, rd, rs, res = _run(gpu_api, 200000, 20, 2000, 30.0, 15, 50, 0.002, seed=3)
    _check_clusters(rd, res, 20, 0.999)
    _check_consensus(sp, res, 0)
