"""GPU: the BASELINE.json configurations that need 8 GPUs, at their PER-GPU shard shapes, plus the strong-scaling (one global set, `--t N`
batches) mode of bench.py with two ranks on one GPU.

C4 = 10 M x 750 bp, 50 species on 8 GPUs -> 1.25 M reads per GPU, abundance_ratio 0.005.
C5 = 2 M x 2 kb CCS, 20 species MIXED (geometric) abundance, k15/w50 on 8 GPUs -> 250 k reads per GPU, abundance_ratio 0.002.
The oracle cannot run these sizes; the checks are the size-independent properties of test_gpu_fullsize.py (pure and complete clusters,
idempotent backward-pointing representative map, every consensus == its generating amplicon).
"""
import os, sys, json, subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
from test_gpu_fullsize import _check_clusters


def _run(gpu_api, n, nsp, L, mu, k, w, ab, seed, abundance=None):
    import torch
    import bench
    from ngspeciesid_amd import pipeline
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.ptable import select_p_table
    dev = torch.device("cuda", 0)
    sp, rd = bench.gen_sorted_reads(gpu_api, n, nsp, L, mu, seed=seed, device=dev, abundance=abundance)
    rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
    res = pipeline.run_hot_path(gpu_api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=k, w=w, abundance_ratio=ab, racon_iter=3,
                                tile_depth=pipeline.TILE_DEPTH, band=0, p_shared=select_p_table(k, w), polish_stop_when_stable=False)
    return sp, rd, rs, res


def _exact(sp, res):
    truths = sorted(s.tobytes().decode() for s in sp)
    got = sorted(c[3] for c in res["centers"])
    assert len(got) == len(truths), "%d consensus sequences for %d species" % (len(got), len(truths))
    bad = [i for i, (g, t) in enumerate(zip(got, truths)) if g != t]
    assert not bad, "%d of %d polished consensus sequences differ from their amplicons" % (len(bad), len(truths))


def test_c4_shard_shape_1p25m_reads_50_species(gpu_api):
    sp, rd, rs, res = _run(gpu_api, 1250000, 50, 750, 17.0, 13, 20, 0.005, seed=7)
    _check_clusters(rd, res, 50, 0.995)
    _exact(sp, res)


def test_c5_shard_shape_250k_ccs_2kb_20_species_geometric_abundance(gpu_api):
    ab = [0.8 ** i for i in range(20)]                       # 20 % of the reads for the most abundant species, 0.29 % (~730 reads) for the rarest
    sp, rd, rs, res = _run(gpu_api, 250000, 20, 2000, 30.0, 15, 50, 0.002, seed=3, abundance=ab)
    _check_clusters(rd, res, 20, 0.999)
    _exact(sp, res)
    sizes = sorted((c[0] for c in res["centers"]), reverse=True)
    assert sizes[0] > 40 * sizes[-1] > 0                      # the abundance really is mixed


@pytest.mark.parametrize("world,lanes_under_ranks", [(2, False), (4, False), (2, True)])
def test_strong_scaling_ranks_on_one_gpu_membership_equals_t_n(world, lanes_under_ranks):
    """bench.py --scaling strong under torch.distributed.run (gloo collectives, HIP compute, all ranks on cuda:0): the merged N-rank membership
    must equal parallelize.tree_cluster(..., N) = the reference's `--t N` schedule on the same global set, and every consensus its amplicon."""
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"; env["NGSID_DIST_BACKEND"] = "gloo"
    for v_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK"): env.pop(v_, None)
    from conftest import release_gpu_memory; release_gpu_memory()          # the ranks below share the GPU with what earlier tests left in THIS process
    if lanes_under_ranks: env["NGSID_LANES_FORCE"] = "1"                   # every rank deals its consensus calls to two contexts, as it does on a GPU of its own (bench.py turns the lanes off when ranks share a device)
    # world 2: `python bench.py --gpus 2 ...` as the driver types it - bench.py starts its own ranks; world 4: under the launcher, as the contract's N > 1 command
    head = [sys.executable] if (world == 2 and not lanes_under_ranks) else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29530 + world + (10 if lanes_under_ranks else 0))]
    cmd = head + [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--reads", "240000", "--scaling", "strong", "--check-membership",
           "--no-cpu-baseline", "--no-extra-step"]
    p = subprocess.run(cmd, env=env, timeout=1200, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, "\n".join(l for l in p.stderr.splitlines() if "Error" in l and "ChildFailed" not in l)[:3000] + "\n...\n" + p.stderr[-1500:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["scaling"] == "strong" and out["config"]["lanes"] == (2 if lanes_under_ranks else 1)
    assert out["rccl_ranks_seen"] == world and out["config"]["ranks"]["ranks_in_the_all_reduce"] == world
    assert set(out["config"]["stage_s_per_step_max_over_ranks"]) >= {"cluster_local", "merge"}
    chk = out["config"]["check"]
    assert chk["membership_equals_reference_t_n"] is True
    assert chk["centers"] == 5 and chk["consensus_edit_distance_vs_truth"] == [0, 0, 0, 0, 0] and chk["cluster_purity"] == 1.0


def test_rccl_collectives_single_rank_strong_scaling():
    """the `nccl` (= RCCL) branch of distributed.TorchComm on the GPU box: the sharded path with its collectives through torch.distributed backend nccl, world size 1
    (NGSID_FORCE_DIST=1 - the test box has one GPU; 2-, 4- and 8-rank runs use gloo).  Payloads travel device -> all_gather_into_tensor -> host as they will on a node."""
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"; env["NGSID_FORCE_DIST"] = "1"; env.pop("NGSID_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29571",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--reads", "200000", "--scaling", "strong", "--check-membership", "--no-cpu-baseline", "--no-extra-step", "--no-cli"]
    p = subprocess.run(cmd, env=env, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    chk = out["config"]["check"]
    assert out["scaling"] == "strong" and chk["membership_equals_reference_t_n"] is True and chk["sharded_consensus_equals_single_process"] is True
    assert chk["centers"] == 5 and chk["consensus_edit_distance_vs_truth"] == [0, 0, 0, 0, 0]


def test_long_ont_reads_5kb_five_species(gpu_api):
    """beyond the BASELINE shapes: 40 k x 5 kb ONT-profile reads (ten polishing windows, amplicon lengths a few bases over a multiple of 500:
    the merged tail window; 128-column first band): every polished consensus == its amplicon"""
    sp, rd, rs, res = _run(gpu_api, 40000, 5, 5000, 17.0, 13, 20, 0.02, seed=3)
    _check_clusters(rd, res, 5, 0.99)
    _exact(sp, res)


@pytest.mark.parametrize("mu", [17.0, 14.0])
def test_mixed_strand_reads_are_merged_and_polished(gpu_api, mu):
    """half of the reads are reverse complements (real ONT data): the reference's flow gives two clusters per species, the rc merge
    (consensus.py:148-183) joins them, and the pooled reads of both strands polish ONE sequence per species: the amplicon or its reverse complement"""
    import torch
    import bench
    from ngspeciesid_amd import pipeline
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.ptable import select_p_table
    dev = torch.device("cuda", 0)
    sp, rd = bench.gen_sorted_reads(gpu_api, 200000, 5, 750, mu, seed=13, device=dev, rc_fraction=0.5)
    rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
    res = pipeline.run_hot_path(gpu_api, rs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), k=13, w=20, abundance_ratio=0.02, racon_iter=3,
                                tile_depth=pipeline.TILE_DEPTH, band=0, p_shared=select_p_table(13, 20), polish_stop_when_stable=False)
    truths = [s.tobytes().decode() for s in sp]
    both = set(truths) | set(pipeline.revcomp_str(t) for t in truths)
    assert len(res["centers"]) == 5, [c[0] for c in res["centers"]]
    assert all(c[3] in both for c in res["centers"]), "a polished consensus is neither an amplicon nor its reverse complement"
    assert all(c[0] > 30000 for c in res["centers"]) and all(len(c[4]) == 2 for c in res["centers"])          # two clusters (fw + rc) behind every centre



@pytest.mark.parametrize("name,total", [("c4", 10000000), ("c5", 2000000)])
def test_c4_c5_composed_eight_shards_on_one_gpu(gpu_api, name, total, compare_single_process=True, out_slots=0):
    """what makes C4 / C5 the 8-GPU configurations, composed AT THEIR STATED SIZES - eight `--t 8` batches of ONE global set (C4: 8 x 1.25 M x 750 bp =
    the 10 M reads of BASELINE.json, 50 species, abundance_ratio 0.005; C5: 8 x 250 k x 2 kb CCS = 2 M reads, 20 species with geometric abundance 0.8^i, k15/w50,
    abundance_ratio 0.002) through distributed.sharded_hot_path: representatives all-gathered and merged by ngsid_merge_representatives, cross-shard abundance cutoff by
    all-reduce, eight weighted partial consensuses per cluster (draft and polished).  Round 5 made C4 fit: the minimizers are a compact CSR (12 bytes per minimizer instead of
    per base), the POA hierarchies run in batches of whole units under a byte budget derived from the free memory and the number of contexts, and the allocator keeps 3 GB of
    headroom for the runtime (eight contexts used to drive the device to "Available Free mem : 100 MB" and an HSA abort).
    The eight ranks are eight threads of this process, each with its own ngsid context on the one GPU (distributed.LocalComm: same payloads,
    exchanged in memory - eight PROCESSES on one MI355X stall in torch's generator kernels before any library call; torch.distributed itself is
    covered by the 2- / 4-process test above and the gloo tests on CPU).
    Checks: membership == parallelize.tree_cluster(.., 8) (= the reference's --t 8 schedule), every polished consensus == its amplicon, pure
    clusters, identical centres on every rank, sharded consensus == the single-process result on the whole set."""
    import torch, time
    import bench
    from ngspeciesid_amd import pipeline, parallelize, distributed, runtime
    from ngspeciesid_amd._capi import ReadSet, cluster_params
    from ngspeciesid_amd.hostutil import make_cluster_fn
    from ngspeciesid_amd.ptable import select_p_table
    import gc
    import ctypes as C0
    _t0 = time.perf_counter()
    def trace(m):
        if os.environ.get("NGSID_TEST_TRACE"): sys.stderr.write("[composed %.1fs free %.1f GB] %s\n" % (time.perf_counter() - _t0, torch.cuda.mem_get_info()[0] / 1e9, m)); sys.stderr.flush()
    gc.collect(); torch.cuda.empty_cache()            # eight contexts' working sets have to fit beside what earlier tests left cached in this process
    gpu_api.set_option("release_scratch", 1)          # (this context and its lane contexts)
    cfg = bench.CONFIGS[name]; world = 8
    K, W, AB, nsp = cfg["k"], cfg["w"], cfg["abundance_ratio"], cfg["species"]
    abundance = [cfg["geometric"] ** i for i in range(nsp)] if cfg["geometric"] else None
    dev = torch.device("cuda", 0)
    sp, rd = bench.gen_sorted_reads(gpu_api, total, nsp, cfg["length"], cfg["mu"], seed=7, device=dev, abundance=abundance, k=K)
    goff = rd["off"]; glens = (goff[1:] - goff[:-1]).cpu().numpy()
    batches = parallelize.batch_list_total_nt(glens, world)
    assert len(batches) == world
    ptab = select_p_table(K, W)
    kw = dict(k=K, w=W, abundance_ratio=AB, racon_iter=3, tile_depth=pipeline.TILE_DEPTH, band=0, p_shared=ptab, polish_stop_when_stable=False)
    shards = []
    for a, b in batches:
        o0, o1 = int(goff[a].item()), int(goff[b].item())
        shards.append(dict(seq=rd["seq"][o0:o1].clone(), qual=rd["qual"][o0:o1].clone(), off=(goff[a:b + 1] - goff[a]).clone(), score=rd["score"][a:b], orig=np.asarray(rd["orig"][a:b], dtype=np.uint32)))
    torch.cuda.synchronize()
    # eight contexts share ONE GPU's memory here: each gets an eighth of the scratch a context takes on a GPU of its own (fewer resident POA tiles
    # and aligner waves - scheduling only, the results do not depend on it)
    import ctypes as C
    small = {"scratch_budget_mb": 3072, "poa_tiles_per_cu": 6}
    if out_slots: small["poa_out_slots"] = out_slots          # (tools/micro/run_composed_full.py: C4 at its full 10 M reads)
    apis = [gpu_api] + [runtime.new_api(0, small) for _ in range(world - 1)]
    for k_, v_ in small.items(): gpu_api.lib.ngsid_ctx_option(gpu_api.ctx, k_.encode(), C.c_int64(v_))
    try:
        def rank_fn(comm):
            s_ = shards[comm.rank]
            rs_ = ReadSet.from_torch(s_["seq"], s_["qual"], s_["off"])
            T = {}
            r = distributed.sharded_hot_path(apis[comm.rank], rs_, s_["score"], acc_rank_local=s_["orig"], comm=comm, timings=T, **kw)
            r["T"] = T
            return r
        t0 = time.perf_counter()
        trace("shards ready, starting the virtual ranks")
        res = distributed.run_virtual_ranks(world, rank_fn)
        dt = time.perf_counter() - t0
        trace("virtual ranks done")
    finally:
        for a_ in apis[1:]: a_.close()
        gpu_api.lib.ngsid_ctx_option(gpu_api.ctx, b"scratch_budget_mb", C.c_int64(32768)); gpu_api.lib.ngsid_ctx_option(gpu_api.ctx, b"poa_tiles_per_cu", C.c_int64(0)); gpu_api.lib.ngsid_ctx_option(gpu_api.ctx, b"poa_out_slots", C.c_int64(4))
    # identical centres on every rank
    cent = [[(c[0], c[1], c[2], c[3]) for c in r["centers"]] for r in res]
    assert all(c == cent[0] for c in cent[1:])
    truths = sorted(s.tobytes().decode() for s in sp)
    assert sorted(c[3] for c in cent[0]) == truths, "%d centres for %d species" % (len(cent[0]), nsp)
    # membership == the reference's --t 8 schedule replayed on one GPU
    starts = [a for a, _ in batches]
    final = np.concatenate([np.asarray([starts[o] + l for o, l in zip(r["final_owner"], r["final_lidx"])], dtype=np.int64) for r in res])
    hrs = ReadSet(rd["seq"].cpu().numpy(), rd["qual"].cpu().numpy(), rd["off"].cpu().numpy().astype(np.uint64))
    fn = make_cluster_fn(gpu_api, hrs, np.asarray(rd["orig"], dtype=np.uint32), cluster_params(k=K, w=W, p_shared=ptab))
    del shards; gc.collect(); torch.cuda.empty_cache()
    trace("replaying --t 8 on one context")
    rep_ref, _, _ = parallelize.tree_cluster(fn, glens, np.asarray(rd["score"]), world)
    trace("replay done")
    assert np.array_equal(final, rep_ref), "sharded membership differs from --t 8 at %d reads" % int((final != rep_ref).sum())
    spc = rd["species"].cpu().numpy()
    big = np.isin(final, np.unique(final)[np.argsort(-np.bincount(np.unique(final, return_inverse=True)[1]))[:nsp]])
    purity = float((spc[final[big]] == spc[big]).mean())
    assert purity > 0.9999 and big.mean() > 0.99, (purity, big.mean())          # (the membership itself is pinned above; a handful of noisy reads join another species' cluster in the reference's --t 8 schedule too)
    # sharded consensus == the single-process path on the whole set (round 5: one context holds the 10 M reads of C4 - 134 GB peak)
    if compare_single_process:
        grs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
        one = pipeline.run_hot_path(gpu_api, grs, rd["score"], acc_rank=np.asarray(rd["orig"], dtype=np.uint32), **kw)
        assert sorted(c[3] for c in one["centers"]) == sorted(c[3] for c in cent[0])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(config=name, total_reads=int(len(glens)), shards=world, species=nsp, wall_s_eight_shards_sharing_one_gpu=round(dt, 2), stage_s_rank0={k_: round(v, 3) for k_, v in res[0]["T"].items()},
                   centres=len(cent[0]), purity_of_the_large_clusters=purity, membership_equals_t8=True, consensus_equals_amplicons=True, equals_single_process=(True if compare_single_process else None)),
              open(os.path.join(ROOT, "gpurun_out", "composed_%s_8_shards_one_gpu.json" % name), "w"), indent=1)
