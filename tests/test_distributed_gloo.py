"""CPU, world_size 2, gloo: the sharded path (ngspeciesid_amd.distributed) with the oracle as the C-ABI backend.
Shards = the reference's own `--t 2` batches, so the merged membership must equal the reference's --t 2 result."""
import os, sys, subprocess, tempfile, json
import numpy as np
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
from oracle_lib import load_oracle, GOLD
from ngspeciesid_amd import distributed, parallelize
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.hostutil import acc_rank, subset_reads
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = np.load(os.path.join(GOLD, "cluster_%s.npz" % sys.argv[2]))
rs = ReadSet(g["seq"], g["qual"], g["off"])
lens = np.diff(g["off"].astype(np.int64))
a, b = parallelize.batch_list_total_nt(lens, world)[rank]
idx = np.arange(a, b)
ar = acc_rank([str(x) for x in g["acc"]])
res = distributed.sharded_hot_path(load_oracle(), subset_reads(rs, idx), g["score"][idx], acc_rank_local=ar[idx], k=int(g["k"]), w=int(g["w"]),
                                   p_shared=g["p_table"], abundance_ratio=0.1, racon_iter=1, tile_depth=8, do_consensus=(sys.argv[4] == "1"))
starts = [s for s, e in parallelize.batch_list_total_nt(lens, world)]
final = np.array([starts[o] + l for o, l in zip(res["final_owner"], res["final_lidx"])])
json.dump(dict(a=int(a), b=int(b), final=final.tolist(), centers=[(c[0], c[3]) for c in res["centers"]]), open(os.path.join(sys.argv[3], "r%d.json" % rank), "w"))
dist.destroy_process_group()
'''


def _run(tag, consensus, world=2):
    tmp = tempfile.mkdtemp()
    wf = os.path.join(tmp, "worker.py"); open(wf, "w").write(WORKER)
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", "29517",
           wf, ROOT, tag, tmp, "1" if consensus else "0"]
    subprocess.run(cmd, check=True, env=env, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return [json.load(open(os.path.join(tmp, "r%d.json" % r))) for r in range(world)]


def test_two_rank_membership_equals_reference_t2():
    g = np.load(os.path.join(ROOT, "tests", "golden", "cluster_synth2k_d15.npz"))
    outs = _run("synth2k_d15", consensus=False)
    final = np.full(len(g["t2_rep_of"]), -1)
    for o in outs:
        final[o["a"]:o["b"]] = o["final"]
    assert np.array_equal(final, g["t2_rep_of"])              # identical cluster membership to the reference's --t 2


def test_two_rank_consensus_agrees_between_ranks():
    outs = _run("synth2k_d15", consensus=True)
    assert outs[0]["centers"] == outs[1]["centers"] and len(outs[0]["centers"]) == 5
    sizes = sorted(c[0] for c in outs[0]["centers"])
    assert sizes == [372, 393, 403, 411, 414]                 # the reference's cluster sizes for this fixture
    # against the truth and against the single-process path on the same reads (oracle backend both times)
    from util_seq import edit_distance
    from oracle_lib import load_oracle
    from ngspeciesid_amd import synth, pipeline
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.hostutil import acc_rank
    truths = [t.tobytes().decode() for t in synth.make_species(5, 750, 0.15, seed=11)]          # oracle/make_golden.py builds the fixture from these
    g = np.load(os.path.join(ROOT, "tests", "golden", "cluster_synth2k_d15.npz"))
    for n, seq in outs[0]["centers"]:
        assert min(edit_distance(seq, t) for t in truths) == 0, "sharded consensus of the %d-read cluster differs from its amplicon" % n
    one = pipeline.run_hot_path(load_oracle(), ReadSet(g["seq"], g["qual"], g["off"]), g["score"], acc_rank=acc_rank([str(x) for x in g["acc"]]), k=int(g["k"]), w=int(g["w"]),
                                p_shared=g["p_table"], abundance_ratio=0.1, racon_iter=1, tile_depth=8)
    assert sorted(c[3] for c in one["centers"]) == sorted(seq for n, seq in outs[0]["centers"]), "sharded and single-process consensus differ"


def test_four_rank_gloo_membership_equals_reference_t4():
    """four gloo processes: merged membership == the reference's --t 4 result (the fixture that holds one: the 10 %-divergence
    "hard" set, where every cross-species read reaches the aligner)"""
    g = np.load(os.path.join(ROOT, "tests", "golden", "cluster_synth600_d10_q14.npz"))
    outs = _run("synth600_d10_q14", consensus=False, world=4)
    final = np.full(len(g["t4_rep_of"]), -1)
    for o in outs:
        final[o["a"]:o["b"]] = o["final"]
    assert np.array_equal(final, g["t4_rep_of"])


def test_eight_rank_gloo_membership_and_consensus_equal_reference_t8():
    """eight gloo PROCESSES through TorchComm (the collectives had run with 2 and 4 ranks): merged membership == the reference's --t 8 result,
    every rank reports the same centres, consensus == amplicons"""
    from util_seq import edit_distance
    from ngspeciesid_amd import synth
    g = np.load(os.path.join(ROOT, "tests", "golden", "cluster_synth2k_d15.npz"))
    outs = _run("synth2k_d15", consensus=True, world=8)
    final = np.full(len(g["t8_rep_of"]), -1)
    for o in outs:
        final[o["a"]:o["b"]] = o["final"]
    assert np.array_equal(final, g["t8_rep_of"])
    assert all(o["centers"] == outs[0]["centers"] for o in outs[1:]) and len(outs[0]["centers"]) == 5
    truths = [t.tobytes().decode() for t in synth.make_species(5, 750, 0.15, seed=11)]
    for n, seq in outs[0]["centers"]:
        assert min(edit_distance(seq, t) for t in truths) == 0


@pytest.mark.parametrize("world,tag", [(8, "synth2k_d15"), (2, "synth2k_d15"), (4, "synth600_d10_q14")])
def test_virtual_ranks_membership_and_consensus(oracle, world, tag):
    """distributed.LocalComm (N virtual ranks = N threads of one process, what the one-GPU emulation of the 8-GPU configurations uses): the same
    sharded_hot_path, collectives in memory.  Membership == the reference's --t N; every rank reports the same centres; consensus == amplicons."""
    from util_seq import edit_distance
    from ngspeciesid_amd import distributed, parallelize, synth
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.hostutil import acc_rank, subset_reads
    g = np.load(os.path.join(ROOT, "tests", "golden", "cluster_%s.npz" % tag))
    rs = ReadSet(g["seq"], g["qual"], g["off"]); lens = np.diff(g["off"].astype(np.int64))
    batches = parallelize.batch_list_total_nt(lens, world)
    ar = acc_rank([str(x) for x in g["acc"]])
    def fn(comm):
        a, b = batches[comm.rank]; idx = np.arange(a, b)
        return distributed.sharded_hot_path(oracle, subset_reads(rs, idx), g["score"][idx], acc_rank_local=ar[idx], k=int(g["k"]), w=int(g["w"]), p_shared=g["p_table"],
                                            abundance_ratio=0.1, racon_iter=1, tile_depth=6, comm=comm)
    res = distributed.run_virtual_ranks(world, fn)
    starts = [s for s, e in batches]
    final = np.concatenate([np.array([starts[o] + l for o, l in zip(r["final_owner"], r["final_lidx"])], dtype=np.int64) for r in res])
    assert np.array_equal(final, g["t%d_rep_of" % world])
    cent = [[(c[0], c[3]) for c in r["centers"]] for r in res]
    assert all(c == cent[0] for c in cent[1:])
    if tag != "synth2k_d15":
        return
    assert len(cent[0]) == 5
    truths = [t.tobytes().decode() for t in synth.make_species(5, 750, 0.15, seed=11)]
    for n, seq in cent[0]:
        assert min(edit_distance(seq, t) for t in truths) == 0


def test_virtual_rank_failure_does_not_hang(oracle):
    from ngspeciesid_amd import distributed
    def fn(comm):
        if comm.rank == 1: raise ValueError("boom")
        return comm.all_gather_obj(dict(x=np.arange(3)))
    with pytest.raises(ValueError):
        distributed.run_virtual_ranks(3, fn)


def test_skewed_shards_weigh_their_share(oracle, monkeypatch):
    """one shard holds 0.5 % of a cluster's reads (the common case for rare species of a skewed sample).  The per-shard partial
    consensuses are merged with the NUMBER OF READS each stands for as weight (ngsid_poa_consensus_weighted; the 1 .. 93 quality-character clamp of
    round 3 gave such a shard 1/93 instead of its share): the weights handed to the library are the read counts, and the sharded result equals the
    single-process result and the amplicons."""
    from ngspeciesid_amd import distributed, synth, pipeline
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.hostutil import subset_reads
    from ngspeciesid_amd.ptable import select_p_table
    sp = synth.make_species(2, 600, 0.15, seed=41)
    rd = synth.make_reads(sp, 1600, mu=15.0, seed=43)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    score, err, keep = oracle.score_reads(rs, 13, 7.0)
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    spc = rd["species"].numpy()[idx]
    a_reads = idx[spc == 0]; b_reads = idx[spc == 1]
    na = len(a_reads); tiny = max(3, na // 200)                       # 0.5 % of species A ...
    half = (na - tiny) // 2
    shards = [np.sort(np.concatenate([a_reads[:half], b_reads[: len(b_reads) // 3]])), np.sort(np.concatenate([a_reads[half:na - tiny], b_reads[len(b_reads) // 3: 2 * len(b_reads) // 3]])),
              np.sort(np.concatenate([a_reads[na - tiny:], b_reads[2 * len(b_reads) // 3:]]))]          # ... sits in the third shard, next to a third of species B
    order = {int(r): x for x, r in enumerate(idx)}
    shards = [np.array(sorted(s.tolist(), key=lambda r: order[int(r)])) for s in shards]             # every shard in score order (the greedy order)
    seen = []
    real = oracle.poa_consensus_weighted
    monkeypatch.setattr(oracle, "poa_consensus_weighted", lambda rset, grp, prm, weight, **kw: (seen.append(np.asarray(weight).tolist()), real(rset, grp, prm, weight, **kw))[1], raising=False)
    kw = dict(k=13, w=20, p_shared=select_p_table(13, 20), abundance_ratio=0.05, racon_iter=1, tile_depth=6)
    def fn(comm):
        s = shards[comm.rank]
        return distributed.sharded_hot_path(oracle, subset_reads(rs, s), score[s], acc_rank_local=np.array([order[int(r)] for r in s], dtype=np.uint32), comm=comm, **kw)
    res = distributed.run_virtual_ranks(3, fn)
    cent = [sorted((c[0], c[3]) for c in r["centers"]) for r in res]
    assert all(c == cent[0] for c in cent[1:]) and len(cent[0]) == 2
    assert any(min(wl) <= tiny and max(wl) >= 100 * min(wl) for wl in seen), "no merge saw the skewed weights: %s" % seen[:4]       # real counts, not 1 .. 93
    truths = sorted(t.tobytes().decode() for t in sp)
    assert sorted(c[1] for c in cent[0]) == truths
    one = pipeline.run_hot_path(oracle, subset_reads(rs, idx), score[idx], acc_rank=np.arange(len(idx), dtype=np.uint32), **kw)
    assert sorted(c[3] for c in one["centers"]) == sorted(c[1] for c in cent[0])
