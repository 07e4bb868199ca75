"""(f3) strand-aware clustering as an option (ngspeciesid_amd/strand.py): on mixed-strand reads the default flow gives two clusters per amplicon
(the reference's behaviour, joined after the drafts by detect_reverse_complements); with the option the clusters are joined at the clustering
level - ONE cluster per amplicon, one draft per amplicon, exact consensus - and the default output is untouched."""
import os
import numpy as np
import pytest
from ngspeciesid_amd import synth, pipeline, strand
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.hostutil import subset_reads
from ngspeciesid_amd.ptable import select_p_table


def _mixed(n=900, nsp=3, L=420, seed=5, mu=17.0):
    sp = synth.make_species(nsp, L, 0.15, seed=seed)
    rd = synth.make_reads(sp, n, mu=mu, seed=seed + 1, rc_fraction=0.5)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    return sp, rd, rs


def _sorted(api, rs, rd):
    score, err, keep = api.score_reads(rs, 13, 7.0)
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    return subset_reads(rs, idx), score[idx], rd["species"].numpy()[idx], rd["strand"].numpy()[idx]


def _run(api, sub, score, **kw):
    return pipeline.run_hot_path(api, sub, score, acc_rank=np.arange(sub.n, dtype=np.uint32), k=13, w=20, abundance_ratio=0.05, racon_iter=2, p_shared=select_p_table(13, 20), **kw)


def _check(sp, spc, strd, off_res, on_res):
    nsp = len(sp)
    truths = [s.tobytes().decode() for s in sp]
    both = set(truths) | set(pipeline.revcomp_str(t) for t in truths)
    # default: two large clusters per species (one per strand), centres joined after the drafts
    big_off = [r for r, c in zip(*np.unique(off_res["rep_of"], return_counts=True)) if c >= 30]
    assert len(big_off) == 2 * nsp
    assert len(off_res["centers"]) == nsp and all(len(c[4]) == 2 for c in off_res["centers"])
    # option: one large cluster per species, holding both strands; the flipped reads are exactly those of the strand opposite to the representative
    rep = on_res["rep_of"]; flip = on_res["flip"]
    big_on = [r for r, c in zip(*np.unique(rep, return_counts=True)) if c >= 30]
    assert len(big_on) == nsp
    for r in big_on:
        mem = rep == r
        assert len(np.unique(spc[mem])) == 1 and set(np.unique(strd[mem]).tolist()) == {0, 1}
        assert np.array_equal(flip[mem], strd[mem] != strd[r])
    assert np.array_equal(rep[rep], rep) and np.all(rep <= np.arange(len(rep)))
    assert len(on_res["centers"]) == nsp and all(len(c[4]) == 1 for c in on_res["centers"])
    assert all(c[3] in both for c in on_res["centers"]) and all(c[3] in both for c in off_res["centers"])
    assert sorted(c[0] for c in on_res["centers"]) == sorted(c[0] for c in off_res["centers"])          # same reads behind every consensus


def test_strand_aware_option_oracle(oracle):
    sp, rd, rs = _mixed()
    sub, score, spc, strd = _sorted(oracle, rs, rd)
    off = _run(oracle, sub, score); on = _run(oracle, sub, score, strand_aware=True)
    assert "flip" not in off
    _check(sp, spc, strd, off, on)
    # single-strand data: nothing to join, membership and consensus as without the option
    sp1 = synth.make_species(2, 420, 0.15, seed=9); rd1 = synth.make_reads(sp1, 300, mu=17.0, seed=10)
    rs1 = ReadSet(rd1["seq"].numpy(), rd1["qual"].numpy(), rd1["off"].numpy().astype(np.uint64))
    sub1, score1, _, _ = _sorted(oracle, rs1, rd1)
    a = _run(oracle, sub1, score1); b = _run(oracle, sub1, score1, strand_aware=True)
    assert np.array_equal(a["rep_of"], b["rep_of"]) and not b["flip"].any() and [c[3] for c in a["centers"]] == [c[3] for c in b["centers"]]


def test_orient_reads_host_equals_per_read_reverse_complement():
    sp, rd, rs = _mixed(n=200)
    flip = np.arange(rs.n) % 3 == 0
    o = strand.orient_reads(rs, flip)
    for i in range(rs.n):
        s, q = rs.get(i)
        assert o.get(i) == ((pipeline.revcomp_str(s), q[::-1]) if flip[i] else (s, q))


@pytest.mark.gpu
def test_strand_aware_option_hip_equals_oracle(gpu_api, oracle):
    sp, rd, rs = _mixed(n=3000, nsp=4, L=750)
    sub, score, spc, strd = _sorted(gpu_api, rs, rd)
    on = _run(gpu_api, sub, score, strand_aware=True); off = _run(gpu_api, sub, score)
    _check(sp, spc, strd, off, on)
    ref = _run(oracle, sub, score, strand_aware=True)
    assert np.array_equal(on["rep_of"], ref["rep_of"]) and np.array_equal(on["flip"], ref["flip"])
    assert [(c[0], c[1], c[2], c[3]) for c in on["centers"]] == [(c[0], c[1], c[2], c[3]) for c in ref["centers"]]
    # device-resident reads (torch tensors, as bench.py holds them): same result, oriented on the device
    import torch
    dev = torch.device("cuda", 0)
    drs = ReadSet.from_torch(torch.from_numpy(sub.seq).to(dev), torch.from_numpy(sub.qual).to(dev), torch.from_numpy(sub.off.astype(np.int64)).to(dev))
    d = _run(gpu_api, drs, score, strand_aware=True)
    assert np.array_equal(d["rep_of"], on["rep_of"]) and [c[3] for c in d["centers"]] == [c[3] for c in on["centers"]]
    # a read set uploaded through the C-ABI (ngsid_reads_upload: no torch tensors behind it): oriented from its host copy, same result
    u = _run(gpu_api, gpu_api.upload_reads(sub), score, strand_aware=True)
    assert np.array_equal(u["rep_of"], on["rep_of"]) and np.array_equal(u["flip"], on["flip"]) and [c[3] for c in u["centers"]] == [c[3] for c in on["centers"]]


@pytest.mark.gpu
def test_cli_strand_aware_flag(gpu_api, tmp_path):
    """`--strand_aware`: final_clusters.tsv holds one large cluster per amplicon; without the flag the files are what they were."""
    from ngspeciesid_amd import fastio
    from ngspeciesid_amd.cli import cli
    sp, rd, rs = _mixed(n=4000, nsp=3, L=750, seed=21)
    spc = rd["species"].numpy(); strd = rd["strand"].numpy()
    names = fastio.Names.from_list(["r%d_sp%d_st%d" % (i, spc[i], strd[i]) for i in range(rs.n)])
    fq = str(tmp_path / "in.fastq"); fastio.write_fastq(fq, np.arange(rs.n), names, rs)
    outs = {}
    for tag, extra in (("off", []), ("on", ["--strand_aware"]), ("off2", [])):
        out = str(tmp_path / tag)
        cli(["--ont", "--fastq", fq, "--outfolder", out, "--t", "1", "--consensus", "--racon", "--racon_iter", "2", "--abundance_ratio", "0.05"] + extra)
        cl = {}
        for line in open(os.path.join(out, "final_clusters.tsv")):
            cid, acc = line.rstrip("\n").split("\t"); cl.setdefault(int(cid), []).append(acc)
        cons = sorted(open(os.path.join(out, d, "consensus.fasta")).read().split("\n")[1] for d in os.listdir(out) if d.startswith("racon_cl_id_"))
        outs[tag] = (cl, cons, open(os.path.join(out, "final_clusters.tsv"), "rb").read())
    assert outs["off"][2] == outs["off2"][2]
    big = lambda cl: [v for v in cl.values() if len(v) >= 100]
    assert len(big(outs["off"][0])) == 6 and len(big(outs["on"][0])) == 3
    for v in big(outs["on"][0]):
        assert len(set(a.split("_")[1] for a in v)) == 1 and len(set(a.split("_")[2] for a in v)) == 2          # one species, both strands
    truths = [s.tobytes().decode() for s in sp]; both = set(truths) | set(pipeline.revcomp_str(t) for t in truths)
    assert len(outs["on"][1]) == 3 and all(c in both for c in outs["on"][1]) and all(c in both for c in outs["off"][1])
