"""How many clustering decisions depend on WHICH co-optimal alignment the aligner returns (parasail 1.2.4's tie-breaks are unpinned).

The oracle's aligner is switched to the other plausible tie-break orders (oracle/ngsid_oracle.c: g_sg_tiebreak) and the greedy clustering of the golden read
sets (the six sets of tests/golden incl. the 10 %-divergence set, plus a 15 %-divergence noisy set made here) is repeated: membership flips (reads whose
representative changes), changes of the counters (mapped_passed, aln_passed, aln_called), and - per aligner call - how many window ratios move and how many
of them sit within 0.05 of the acceptance threshold 0.4.      python tools/r04_tiebreak_envelope.py --out profiles/r04_tiebreak_envelope.json
"""
import argparse, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle_lib import load_oracle, GOLD
from ngspeciesid_amd import synth
from ngspeciesid_amd._capi import ReadSet, cluster_params
from ngspeciesid_amd.ptable import select_p_table
from ngspeciesid_amd.hostutil import subset_reads
from test_oracle_golden import _acc_rank

MODES = {0: "this build: diag > E > F, extension on ties, first maximum", 1: "diag > F > E", 2: "E / F prefer opening on ties", 4: "last maximum as end cell",
         8: "E > F > diag (gaps first)", 9: "F > E > diag", 3: "diag > F > E + opening", 7: "diag > F > E + opening + last maximum", 15: "F > E > diag + opening + last maximum"}


def sets(orc):
    for tag in ("sample_h1", "synth2k_d15", "synth600_d10_q14", "synth300_ccs", "synth1200_k25", "synth1200_k30"):
        g = np.load(os.path.join(GOLD, "cluster_%s.npz" % tag), allow_pickle=False)
        yield tag, ReadSet(g["seq"], g["qual"], g["off"]), cluster_params(k=int(g["k"]), w=int(g["w"]), p_shared=g["p_table"]), _acc_rank([str(a) for a in g["acc"]])
    # noisy 15 %-divergence set (mu = 12: ~12 % read error, the regime in which the alignment stage decides most memberships)
    sp = synth.make_species(6, 750, 0.15, seed=3)
    rd = synth.make_reads(sp, 3000, mu=12.0, seed=17)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    score, err, keep = orc.score_reads(rs, 13, 7.0)
    idx = np.nonzero(keep)[0]; idx = idx[np.argsort(-score[idx], kind="stable")]
    yield "synth3000_d15_mu12 (made here)", subset_reads(rs, idx), cluster_params(k=13, w=20, p_shared=select_p_table(13, 20)), np.arange(len(idx), dtype=np.uint32)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--out", default=None); ap.add_argument("--pairs-only", action="store_true"); a = ap.parse_args()
    orc = load_oracle()
    tb = lambda m: orc.lib.ongsid_debug_sg_tiebreak(C.c_int32(m))
    rec = {"_what": __doc__.split("\n\n")[0], "modes": {str(k): v for k, v in MODES.items()}, "sets": []}
    for tag, rs, prm, acc in ([] if a.pairs_only else sets(orc)):
        base = None; row = {"set": tag, "reads": int(rs.n), "by_mode": {}}
        for m in MODES:
            old = tb(m)
            try:
                rep, herr, st, cnt = orc.cluster_greedy(rs, prm, acc_rank=acc)
            finally:
                tb(old)
            if m == 0:
                base = (rep.copy(), [int(c) for c in cnt[:4]]); row["counters_mode0"] = dict(zip(("mapped_passed", "aln_passed", "aln_called", "new_representatives"), base[1]))
            row["by_mode"][str(m)] = {"membership_flips": int((rep != base[0]).sum()), "counter_deltas": [int(c) - b for c, b in zip(cnt[:4], base[1])]}
        rec["sets"].append(row); print(json.dumps(row), flush=True)
    # ---- per-pair view: how far does the window ratio (cluster.py:146-167) move, and how many pairs sit that close to the acceptance threshold 0.4
    sp = synth.make_species(6, 750, 0.15, seed=3)
    rd = synth.make_reads(sp, 1500, mu=12.0, seed=23)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    spc = rd["species"].numpy(); rng = np.random.default_rng(5)
    qi = np.arange(rs.n, dtype=np.uint32)
    same = np.array([rng.choice(np.nonzero(spc == spc[i])[0]) for i in range(rs.n)], dtype=np.uint32)          # a read of the same species ...
    other = np.array([rng.choice(np.nonzero(spc != spc[i])[0]) for i in range(rs.n)], dtype=np.uint32)        # ... and one of another species (15 % divergence)
    qidx = np.concatenate([qi, qi]); tidx = np.concatenate([same, other]); qlen = np.diff(rs.off.astype(np.int64))[qidx]
    opens = np.full(len(qidx), 2, dtype=np.int32); mid = np.full(len(qidx), 10, dtype=np.int32)
    pairs = {"pairs": int(len(qidx)), "what": "1 500 reads at mu = 12 (~12 %% read error) against a read of the same species and against one of another species (15 %% divergence), raw sequences, open 2, match_id 10 of k 13", "by_mode": {}}
    base = None
    for m in MODES:
        old = tb(m)
        try: r = orc.sg_align_batch(rs, rs, qidx, tidx, opens, 1, 2, -2, 13, mid)
        finally: tb(old)
        ratio = r[3] / qlen.astype(np.float64)
        if m == 0: base = (r[0].copy(), ratio.copy()); pairs["ratio_quantiles_mode0_same_species"] = [round(float(x), 3) for x in np.quantile(ratio[:rs.n], [0, 0.01, 0.5, 1])]; pairs["ratio_quantiles_mode0_other_species"] = [round(float(x), 3) for x in np.quantile(ratio[rs.n:], [0, 0.5, 0.99, 1])]
        d = np.abs(ratio - base[1])
        pairs["by_mode"][str(m)] = {"scores_equal": bool(np.array_equal(r[0], base[0])), "pairs_whose_ratio_moves": int((d > 0).sum()), "max_abs_ratio_shift": round(float(d.max()), 4),
                                    "decisions_that_flip_at_0.4": int(((ratio >= 0.4) != (base[1] >= 0.4)).sum())}
    mx = max(v["max_abs_ratio_shift"] for v in pairs["by_mode"].values())
    pairs["pairs_within_max_shift_of_0.4"] = int((np.abs(base[1] - 0.4) <= mx).sum())
    rec["pairs"] = pairs; print(json.dumps(pairs), flush=True)
    rec["max_membership_flips"] = max([v["membership_flips"] for s in rec["sets"] for v in s["by_mode"].values()] or [0])
    if a.out: json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
