"""The bench's read generators (ngspeciesid_amd/synth.py): the counter-based generator used by multi-process runs (no torch.Generator) draws the same
error profile as the torch generator the one-GPU workloads keep, is deterministic, and its reads cluster by species."""
import numpy as np
import torch
from ngspeciesid_amd import synth
from ngspeciesid_amd._capi import ReadSet, cluster_params
from ngspeciesid_amd.ptable import select_p_table


def test_hash_generator_matches_the_profile_of_the_torch_generator(oracle):
    sp = synth.make_species(3, 500, 0.15, seed=2)
    kw = dict(mu=17.0, seed=4, abundance=[0.5, 0.3, 0.2], rc_fraction=0.25)
    a = synth.make_reads(sp, 4000, rng="hash", **kw); b = synth.make_reads(sp, 4000, rng="hash", **kw); t = synth.make_reads(sp, 4000, **kw)
    assert torch.equal(a["seq"], b["seq"]) and torch.equal(a["qual"], b["qual"]) and torch.equal(a["off"], b["off"])          # deterministic
    c = synth.make_reads(sp, 4000, rng="hash", **dict(kw, seed=5))
    assert not torch.equal(a["seq"][:1000], c["seq"][:1000])                                                                  # the seed matters
    def stats(r):
        q = r["qual"].numpy().astype(np.float64) - 33
        return np.diff(r["off"].numpy()).mean(), q.mean(), q.std(), (10.0 ** (-q / 10.0)).mean(), r["strand"].float().mean().item(), np.bincount(r["species"].numpy(), minlength=3) / 4000.0
    sa, st = stats(a), stats(t)
    assert abs(sa[0] - st[0]) < 1.0 and abs(sa[1] - st[1]) < 0.2 and abs(sa[2] - st[2]) < 0.1 and abs(sa[3] - st[3]) < 0.003 and abs(sa[4] - 0.25) < 0.03
    assert np.all(np.abs(sa[5] - np.array([0.5, 0.3, 0.2])) < 0.03)
    # forward-strand reads of the hash generator cluster by species (oracle backend)
    f = synth.make_reads(sp, 600, mu=17.0, seed=9, rng="hash")
    rs = ReadSet(f["seq"].numpy(), f["qual"].numpy(), f["off"].numpy().astype(np.uint64))
    score, err, keep = oracle.score_reads(rs, 13, 7.0)
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    from ngspeciesid_amd.hostutil import subset_reads
    rep, _, _, _ = oracle.cluster_greedy(subset_reads(rs, idx), cluster_params(k=13, w=20, p_shared=select_p_table(13, 20)))
    spc = f["species"].numpy()[idx]
    big = [r for r in np.unique(rep) if (rep == r).sum() > 50]
    assert len(big) == 3 and all(len(np.unique(spc[rep == r])) == 1 for r in big)
