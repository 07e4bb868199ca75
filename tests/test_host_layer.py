"""CPU: the reference-shaped host layer (cluster.reads_to_clusters / parallelize.parallel_clustering dict API) with the oracle as
the C-ABI backend, against the reference's own clusters dicts (tests/golden)."""
import os, argparse
import numpy as np
import pytest
from oracle_lib import GOLD
from ngspeciesid_amd import cluster, parallelize
from ngspeciesid_amd.ptable import p_emp_probs_dict


def ref_args(**kw):
    a = argparse.Namespace(k=13, w=20, min_shared=5, mapped_threshold=0.7, aligned_threshold=0.4, symmetric_map_align_thresholds=False,
                           min_fraction=0.8, min_prob_no_hits=0.1, print_output=10 ** 9, nr_cores=1, batch_type="total_nt")
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def load_read_array(tag):
    g = np.load(os.path.join(GOLD, "cluster_%s.npz" % tag), allow_pickle=False)
    off = g["off"].astype(np.int64)
    ra = []
    for i in range(len(off) - 1):
        ra.append((i, 0, str(g["acc"][i]), g["seq"][off[i]:off[i + 1]].tobytes().decode(), g["qual"][off[i]:off[i + 1]].tobytes().decode(), float(g["score"][i])))
    return g, ra


@pytest.mark.parametrize("tag,t", [("sample_h1", 1), ("sample_h1", 8), ("synth600_d10_q14", 4)])
def test_dict_api_matches_reference(oracle, tag, t):
    g, ra = load_read_array(tag)
    args = ref_args(k=int(g["k"]), w=int(g["w"]), nr_cores=t)
    pt = p_emp_probs_dict(args.k, args.w)
    if t == 1:
        clusters = {i: [acc] for i, _, acc, *_ in ra}; reps = {i: (i, b, acc, s, q, sc) for i, b, acc, s, q, sc in ra}
        res = cluster.reads_to_clusters(clusters, reps, ra, pt, {}, 1, args, api=oracle)
        clusters, reps, _, _ = res[1]
    else:
        clusters, reps = parallelize.parallel_clustering(ra, pt, args, api=oracle)
    keys, off, order = g["t%d_cl_keys" % t], g["t%d_cl_off" % t], g["t%d_cl_order" % t]
    assert sorted(clusters.keys()) == sorted(int(k) for k in keys)
    for i, kk in enumerate(keys):
        assert clusters[int(kk)] == [str(g["acc"][j]) for j in order[off[i]:off[i + 1]]]
    for kk in keys:                                       # representatives carry the 8-tuple with the HPC error rate
        tup = reps[int(kk)]
        if len(tup) == 8:
            ref = g["t%d_rep_err" % t][int(kk)]
            assert abs(tup[6] - ref) <= 16 * np.spacing(ref)


def test_clusters_from_rep_matches_the_plain_definition():
    """pipeline.clusters_from_rep (radix-friendly grouping) == stable argsort + np.unique on the representative map, incl. > 65 535 clusters"""
    import numpy as np
    from ngspeciesid_amd.pipeline import clusters_from_rep, select_centers
    rng = np.random.default_rng(5)
    for n, nrep in ((1, 1), (50, 7), (20000, 300), (150000, 70000)):
        reps = np.sort(rng.choice(n, nrep, replace=False)); reps[0] = 0
        rep_of = reps[rng.integers(0, nrep, n)].astype(np.int32)
        rep_of = np.minimum(rep_of, np.arange(n, dtype=np.int32))          # a read never joins a later read ...
        rep_of[reps] = reps                                                # ... and representatives are their own
        rep_of = rep_of[rep_of]; rep_of = rep_of[rep_of]                   # make the map idempotent
        keep = rep_of[rep_of] == rep_of
        assert keep.all()
        r, order, goff, counts = clusters_from_rep(rep_of)                  # int32 map (what the clustering call returns): ngsid_host_group_by_rep32
        r64, order64, goff64, counts64 = clusters_from_rep(rep_of.astype(np.int64))
        assert np.array_equal(r, r64) and np.array_equal(order, order64) and np.array_equal(goff, goff64) and np.array_equal(counts, counts64)
        o2 = np.argsort(rep_of, kind="stable").astype(np.uint32)
        r2, st, c2 = np.unique(rep_of[o2], return_index=True, return_counts=True)
        assert np.array_equal(order, o2) and np.array_equal(r, r2) and np.array_equal(counts, c2) and np.array_equal(goff[:-1], st) and goff[-1] == n
        score = rng.random(n)
        sel = select_centers(r, counts, score, 3)
        assert all(counts[i] >= 3 for i in sel)
        assert sel == sorted(sel, key=lambda i: (-counts[i], -score[r[i]]))


def test_write_fastq_subcommand(tmp_path):
    """the reference's `write_fastq` sub-command (NGSpeciesID:161-182,238-245): <cluster id>.fastq per cluster with at least --N reads, looked up by whole header line"""
    import os, pytest
    from ngspeciesid_amd import cli
    fq = tmp_path / "r.fastq"
    fq.write_text("".join("@r%d\n%s\n+\n%s\n" % (i, "ACGT" * (i + 1), "I" * (4 * (i + 1))) for i in range(5)))
    cl = tmp_path / "final_clusters.tsv"
    cl.write_text("0\tr0\n0\tr3\n1\tr1\n2\tr2\n0\tr4\n")
    out = tmp_path / "o"
    with pytest.raises(SystemExit) as e:
        cli.cli(["--fastq", str(fq), "write_fastq", "--clusters", str(cl), "--fastq", str(fq), "--outfolder", str(out), "--N", "2"])
    assert e.value.code == 0
    assert sorted(os.listdir(out)) == ["0.fastq"]
    assert (out / "0.fastq").read_text() == "@r0\nACGT\n+\nIIII\n@r3\n%s\n+\n%s\n@r4\n%s\n+\n%s\n" % ("ACGT" * 4, "I" * 16, "ACGT" * 5, "I" * 20)
    with pytest.raises(SystemExit):
        cli.cli(["--fastq", str(fq), "write_fastq", "--clusters", str(cl), "--fastq", str(fq), "--outfolder", str(out), "--N", "0"])
    assert sorted(os.listdir(out)) == ["0.fastq", "1.fastq", "2.fastq"]
