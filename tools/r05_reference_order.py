"""Round 5: the REFERENCE-ORDER mode of the polisher (one graph per window in first-position order, no trimming of NGS windows, untrimmed draft) and
the racon rules this build replaces - measured on the CPU oracle, per polishing iteration, against the generating amplicons.

Rules (oracle-only switches, oracle/ngsid_oracle_poa.c: ongsid_debug_polish_rules; the HIP kernels implement rules = 0):
  bit 0  overlap-span clipping of the read -> backbone alignment (minimap2's q_begin / q_end before racon's edlib call): only the columns between the first and the last
         run of >= 15 equal columns are kept
  bit 1  layers that do not span their window are aligned globally to the SUB-graph of their span (racon src/window.cpp) instead of end-free to the whole graph
  bit 2  PROBE (not a racon rule): unaligned head / tail bases of a window layer create no nodes - isolates the effect of source / sink nodes at the window edges

Data: 5 x 2 000-read C3-shaped clusters (750 bp, 15 % divergence) at mu = 17 and at mu = 14 (the sets of profiles/r04_consensus_deviation.json), each polished from its own
reference-order draft AND from the exact amplicon; and the reference's own test/sample_h1.fastq (shipped mode vs reference-order mode under every rule set).
    python tools/r05_reference_order.py --out profiles/r05_reference_order.json [--jobs 8]
"""
import argparse, json, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

RULESETS = [0, 1, 3, 4, 5, 7]
NAMES = {0: "round-4 restatement", 1: "+ overlap-span clipping", 3: "+ clipping + sub-graph alignment (both racon rules)", 4: "probe only: no source / sink nodes from layer ends",
         5: "clipping + probe", 7: "clipping + sub-graph + probe"}


def one_cluster(job):
    mu, species, nreads = job
    from ngspeciesid_amd import synth, pipeline
    from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, POA_LOCAL
    from ngspeciesid_amd.hostutil import subset_reads
    from oracle_lib import load_oracle
    from util_seq import edit_distance, overlap_distance
    api = load_oracle()
    sp = synth.make_species(5, 750, 0.15, seed=1)
    rd = synth.make_reads(sp, 5 * nreads, mu=mu, seed=11)                         # the read set of the round-4 record; this worker keeps one species of it
    spc = rd["species"].numpy(); off = rd["off"].numpy()
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), off.astype(np.uint64))
    score, err, keep = api.score_reads(rs, 13, 7.0)
    idx = np.nonzero(keep & (spc == species))[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    sub = subset_reads(rs, idx)
    truth = sp[species].tobytes().decode()
    draft = api.poa_consensus(sub, [0, sub.n], poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=0, band=0, node_cap=160, trim=0))[0]
    rec = {"mu": mu, "species": species, "reads": int(sub.n), "draft_vs_truth": edit_distance(draft, truth), "draft_interior_vs_truth": overlap_distance(draft, truth), "rules": {}}
    for r in RULESETS:
        api.lib.ongsid_debug_polish_rules(C.c_int32(r))
        out = {}
        for nm, start in (("from_draft", draft), ("from_truth", truth)):
            prm = polish_params(iters=3, k=13, w=20, tile_depth=0, band=0, node_cap=160, trim=1, stop_when_stable=0)
            its, used = api.polish_trace(ReadSet.from_strings([start]), sub, [0, sub.n], prm)
            out[nm] = {"edits_vs_truth_per_iteration": [edit_distance(s[0], truth) for s in its], "interior_edits_vs_truth_per_iteration": [overlap_distance(s[0], truth) for s in its]}
        rec["rules"][str(r)] = out
    api.lib.ongsid_debug_polish_rules(C.c_int32(0))
    return rec


def sample_h1():
    from ngspeciesid_amd import pipeline
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.ptable import select_p_table
    from ngspeciesid_amd.hostutil import subset_reads
    from oracle_lib import load_oracle
    from util_seq import edit_distance, overlap_distance
    api = load_oracle()
    seqs, quals = [], []
    with open(os.path.join(ROOT, "tests", "golden", "sample_h1.fastq")) as f:
        L = f.read().split("\n")
    for i in range(0, len(L) - 3, 4):
        seqs.append(L[i + 1]); quals.append(L[i + 3])
    rs = ReadSet.from_strings(seqs, quals)
    score, err, keep = api.score_reads(rs, 13, 7.0)
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    sub = subset_reads(rs, idx)

    def run(mode):
        r = pipeline.run_hot_path(api, sub, score[idx], acc_rank=np.arange(sub.n, dtype=np.uint32), k=13, w=20, abundance_ratio=0.1, racon_iter=3, band=0,
                                  p_shared=select_p_table(13, 20), polish_stop_when_stable=False, **mode)
        assert len(r["centers"]) == 1
        return r["centers"][0][3]
    shipped = run(dict(tile_depth=pipeline.TILE_DEPTH, draft_trim=1, polish_trim=2, node_cap=0))
    out = {"data": "test/sample_h1.fastq, merged cluster (13.6 % read error); no truth is known", "shipped_len": len(shipped), "reference_order": {}}
    for r in RULESETS:
        api.lib.ongsid_debug_polish_rules(C.c_int32(r))
        ref = run(dict(tile_depth=0, draft_trim=0, polish_trim=1, node_cap=160))
        out["reference_order"][str(r)] = {"len": len(ref), "edits_vs_shipped": edit_distance(shipped, ref), "interior_edits_vs_shipped": overlap_distance(shipped, ref),
                                          "interior_per_10kb": round(overlap_distance(shipped, ref) * 1e4 / len(ref), 1), "sequence": ref}
    api.lib.ongsid_debug_polish_rules(C.c_int32(0))
    out["shipped"] = shipped
    return out


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--jobs", type=int, default=8); ap.add_argument("--reads", type=int, default=2000)
    a = ap.parse_args()
    import multiprocessing as mp
    jobs = [(mu, s, a.reads) for mu in (17.0, 14.0) for s in range(5)]
    t = time.time()
    with mp.get_context("spawn").Pool(a.jobs) as pool:
        h1 = pool.apply_async(sample_h1)
        clusters = pool.map(one_cluster, jobs, chunksize=1)
        h1 = h1.get()
    summary = {}
    for r in RULESETS:
        fd = [c["rules"][str(r)]["from_draft"]["edits_vs_truth_per_iteration"] for c in clusters]; ft = [c["rules"][str(r)]["from_truth"]["edits_vs_truth_per_iteration"] for c in clusters]
        fdi = [c["rules"][str(r)]["from_draft"]["interior_edits_vs_truth_per_iteration"] for c in clusters]
        summary[str(r)] = {"rules": NAMES[r],
                           "clusters_where_polishing_increases_the_distance_of_the_draft": sum(1 for c, x in zip(clusters, fd) if x[-1] > c["draft_vs_truth"]),
                           "clusters_where_the_exact_amplicon_is_not_a_fixed_point": sum(1 for x in ft if x[-1] > 0),
                           "sum_of_edits_after_3_iterations_from_draft": sum(x[-1] for x in fd), "sum_of_interior_edits_after_3_iterations_from_draft": sum(x[-1] for x in fdi),
                           "sum_of_edits_of_the_drafts": sum(c["draft_vs_truth"] for c in clusters), "sum_of_edits_after_3_iterations_from_truth": sum(x[-1] for x in ft)}
    rec = {"_what": __doc__, "summary_over_10_synthetic_clusters": summary, "sample_h1": h1, "clusters": clusters, "seconds": round(time.time() - t, 1)}
    print(json.dumps(summary, indent=1)); print(json.dumps({k_: {kk: vv for kk, vv in v.items() if kk != "sequence"} for k_, v in h1["reference_order"].items()}, indent=1))
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
