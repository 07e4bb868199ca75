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


@pytest.mark.parametrize("world", [2, 4])
def test_strong_scaling_ranks_on_one_gpu_membership_equals_t_n(world):
    """bench.py --scaling strong under torch.distributed.run (gloo collectives, HIP compute, all ranks on cuda:0): the merged N-rank membership
    must equal parallelize.tree_cluster(..., N) = the reference's `--t N` schedule on the same global set, and every consensus its amplicon."""
    env = dict(os.environ); env["MASTER_ADDR"] = "127.0.0.1"; env["NGSID_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(29530 + world),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--reads", "240000", "--scaling", "strong", "--check-membership",
           "--no-cpu-baseline", "--no-extra-step"]
    p = subprocess.run(cmd, env=env, timeout=1200, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["scaling"] == "strong"
    chk = out["config"]["check"]
    assert chk["membership_equals_reference_t_n"] is True
    assert chk["centers"] == 5 and chk["consensus_edit_distance_vs_truth"] == [0, 0, 0, 0, 0] and chk["cluster_purity"] == 1.0


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
