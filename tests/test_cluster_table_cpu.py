"""CPU: fastpath.cluster_table (round 4: sizes from a bincount, list order from a scatter - no sort over the reads) against the sort-based definition it replaced
(np.unique + lexsort by (cluster, position) / (cluster, -score, position)), on random clusterings incl. singletons, ties in the scores and the --t N list order."""
import numpy as np
import pytest
from ngspeciesid_amd import fastpath


class _SR:
    def __init__(self, n, score): self.n = n; self.score = score


def _definition(sr, sel, rep_of, pos):
    reps, inv, sizes = np.unique(rep_of[sel], return_inverse=True, return_counts=True)
    out_order = np.lexsort((reps, -sr.score[reps], -sizes))
    out_rank = np.empty(len(reps), dtype=np.int64); out_rank[out_order] = np.arange(len(reps))
    cl = out_rank[inv]
    list_order = sel[np.lexsort((pos[sel], cl))]
    file_order = sel[np.lexsort((pos[sel], -sr.score[sel], cl))]
    goff = np.zeros(len(reps) + 1, dtype=np.int64); goff[1:] = np.cumsum(sizes[out_order])
    return reps[out_order], sizes[out_order], goff, list_order, file_order, np.sort(cl)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("single", [True, False])
def test_cluster_table_equals_the_sort_based_definition(seed, single):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 4000))
    score = np.sort(np.round(rng.random(n) * 50, int(rng.integers(0, 3))))[::-1].copy()       # descending, with ties
    sel = np.sort(rng.choice(n, size=int(n * rng.uniform(0.5, 1.0)), replace=False)).astype(np.int64)
    k = int(rng.integers(1, max(2, len(sel) // 3)))
    # representatives: the first read of every cluster (a read joins a representative that comes earlier in the processing order)
    rep_idx = np.sort(rng.choice(len(sel), size=min(k, len(sel)), replace=False)); rep_idx[0] = 0
    assign = np.array([rep_idx[rng.integers(0, np.searchsorted(rep_idx, i, "right"))] for i in range(len(sel))])
    assign[rep_idx] = rep_idx
    rep_of = np.arange(n, dtype=np.int64); rep_of[sel] = sel[assign]
    pos = np.zeros(n, dtype=np.int64)
    for r in np.unique(assign):
        members = np.nonzero(assign == r)[0]
        if single:      # one pass: representative first, then the members in processing order
            order = members
        else:           # --t N: the representative first, the rest in the order the joined lists were appended (any order)
            rest = members[members != r]; rng.shuffle(rest); order = np.concatenate(([r], rest))
        pos[sel[order]] = np.arange(len(order))
    sr = _SR(n, score)
    got = fastpath.cluster_table(sr, sel, rep_of, pos, single_pass=single)
    exp = _definition(sr, sel, rep_of, pos)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)
