"""Deviation of the shipped consensus mode from the mode that restates the reference's tools, POLISHED vs POLISHED.

  shipped          : POA tiles of pipeline.TILE_DEPTH reads, coverage-trimmed tile consensuses (draft trim 1, polish trim 2), 1/3 rule on upper levels
  reference-order  : ONE graph per cluster / window in file order (tile_depth 0), no trimming of the draft (spoa), racon's window rule (trim 1: TGS windows only),
                     graph capacity large enough that no graph is closed early (node_cap 160 = 10 x the first sequence), all iterations

Data: the reference's own test/sample_h1.fastq (253 reads in two strand clusters, merged by detect_reverse_complements) and 5 x 2 000-read C3-shaped clusters
(750 bp, 15 % divergence) at mu = 17 and mu = 14.  Output: edits between the two polished sequences per cluster and per 10 kb, and each mode's distance to
the generating amplicon (synthetic sets).  Backend: --backend hip (GPU box) or oracle (CPU; the HIP path equals the oracle byte for byte, tests/).
    python tools/r04_consensus_deviation.py --backend oracle --out profiles/r04_consensus_deviation.json
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from ngspeciesid_amd import synth, pipeline, fastio
from ngspeciesid_amd._capi import ReadSet
from ngspeciesid_amd.ptable import select_p_table
from ngspeciesid_amd.hostutil import subset_reads
from util_seq import edit_distance, overlap_distance

SHIPPED = dict(tile_depth=pipeline.TILE_DEPTH, draft_trim=1, polish_trim=2, node_cap=0)
REFORDER = dict(tile_depth=0, draft_trim=0, polish_trim=1, node_cap=160)


def best_ed(a, truths):
    return min(min(edit_distance(a, t), edit_distance(pipeline.revcomp_str(a), t)) for t in truths)


def run(api, rs, ab, mode, iters=3):
    score, err, keep = api.score_reads(rs, 13, 7.0)
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    sub = subset_reads(rs, idx)
    r = pipeline.run_hot_path(api, sub, score[idx], acc_rank=np.arange(sub.n, dtype=np.uint32), k=13, w=20, abundance_ratio=ab, racon_iter=iters, band=0,
                              p_shared=select_p_table(13, 20), polish_stop_when_stable=False, **mode)
    return [(c[0], c[2], c[3]) for c in r["centers"]]


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--backend", default="oracle"); ap.add_argument("--out", default=None); ap.add_argument("--reads", type=int, default=2000)
    a = ap.parse_args()
    if a.backend == "hip":
        from ngspeciesid_amd import runtime; api = runtime.get_api(0)
    else:
        from oracle_lib import load_oracle; api = load_oracle()
    rec = {"_what": __doc__.split("\n\n")[0], "backend": a.backend, "shipped": SHIPPED, "reference_order": REFORDER, "sets": []}
    # --- the reference's own reads
    seqs = None
    if seqs is None:
        seqs, quals = [], []
        with open(os.path.join(ROOT, "tests", "golden", "sample_h1.fastq")) as f:
            L = f.read().split("\n")
        for i in range(0, len(L) - 3, 4):
            seqs.append(L[i + 1]); quals.append(L[i + 3])
    rs = ReadSet.from_strings(seqs, quals)
    t = time.time(); A = run(api, rs, 0.1, SHIPPED); B = run(api, rs, 0.1, REFORDER)
    assert len(A) == len(B) == 1
    d_draft = edit_distance(A[0][1], B[0][1]); d_pol = edit_distance(A[0][2], B[0][2]); d_int = overlap_distance(A[0][2], B[0][2])
    rec["sets"].append({"data": "test/sample_h1.fastq (reference's own reads, 13.6 %% error), merged cluster of %d reads" % A[0][0], "polished_len": [len(A[0][2]), len(B[0][2])],
                        "draft_edits": d_draft, "polished_edits": d_pol, "polished_edits_per_10kb": round(d_pol * 10000.0 / len(B[0][2]), 2),
                        "polished_interior_edits": d_int, "polished_interior_edits_per_10kb": round(d_int * 10000.0 / len(B[0][2]), 2), "polished_shipped": A[0][2], "polished_reference_order": B[0][2], "seconds": round(time.time() - t, 1)})
    print(rec["sets"][-1], flush=True)
    # --- C3-shaped clusters
    for mu in (17.0, 14.0):
        sp = synth.make_species(5, 750, 0.15, seed=1)
        rd = synth.make_reads(sp, 5 * a.reads, mu=mu, seed=11)
        rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
        truths = [s.tobytes().decode() for s in sp]
        t = time.time(); A = run(api, rs, 0.02, SHIPPED); B = run(api, rs, 0.02, REFORDER)
        A = sorted(A, key=lambda c: min(range(5), key=lambda i: best_ed(c[2], [truths[i]]))); B = sorted(B, key=lambda c: min(range(5), key=lambda i: best_ed(c[2], [truths[i]])))
        assert len(A) == len(B) == 5, (len(A), len(B))
        per = []
        for x, y in zip(A, B):
            yy = y[2] if edit_distance(x[2], y[2]) <= edit_distance(x[2], pipeline.revcomp_str(y[2])) else pipeline.revcomp_str(y[2])
            d = edit_distance(x[2], yy)
            per.append({"reads": x[0], "polished_edits": d, "polished_interior_edits": overlap_distance(x[2], yy), "reference_order_interior_vs_truth": min(overlap_distance(yy, t) for t in truths) if d else 0, "shipped_vs_truth": best_ed(x[2], truths), "reference_order_vs_truth": best_ed(y[2], truths),
                        "draft_shipped_vs_truth": best_ed(x[1], truths), "draft_reference_order_vs_truth": best_ed(y[1], truths)})
        tot = sum(p["polished_edits"] for p in per); bases = sum(len(y[2]) for y in B); toti = sum(p["polished_interior_edits"] for p in per)
        rec["sets"].append({"data": "synthetic C3 shape: 5 species x %d reads x 750 bp, mu %.0f" % (a.reads, mu), "clusters": per, "polished_edits_total": tot,
                            "polished_edits_per_10kb": round(tot * 10000.0 / bases, 2), "polished_interior_edits_total": toti,
                            "polished_interior_edits_per_10kb": round(toti * 10000.0 / bases, 2), "seconds": round(time.time() - t, 1)})
        print(rec["sets"][-1], flush=True)
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
