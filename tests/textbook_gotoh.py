"""Textbook semi-global affine-gap alignment SCORE (Gotoh 1982, the three-matrix form of Durbin et al., "Biological sequence
analysis", section 2.4), written independently of oracle/ngsid_oracle.c and of the HIP kernels.  Test infrastructure.

It anchors what the reference asks of parasail at cluster.py:131-135 and consensus.py:59-63 as far as documentation pins it
(SURVEY.md 8c): sg = all four end gaps free, a gap of length l costs open + (l - 1) * ext, matrix_create("ACGT", match, mismatch)
(case-insensitive, any other character scores 0).  The optimal SCORE does not depend on traceback tie-breaks, so this is the
part of row a10 that can be checked against something other than this build's own aligner.

    M [i][j]  best score of an alignment of q[:i], t[:j] that ends in the column (q[i-1], t[j-1])
    Ix[i][j]  ... that ends in a gap column consuming q[i-1]   (CIGAR 'I')
    Iy[i][j]  ... that ends in a gap column consuming t[j-1]   (CIGAR 'D')
    free leading gaps: M[i][0] = M[0][j] = 0;   free trailing gaps: answer = max over the last row and the last column
"""
NEG = -10 ** 9


def sub_score(a: str, b: str, match: int, mismatch: int) -> int:
    a = a.upper(); b = b.upper()
    if a not in "ACGT" or b not in "ACGT":
        return 0
    return match if a == b else mismatch


def gotoh_semiglobal_score(q: str, t: str, match=2, mismatch=-2, gap_open=3, gap_ext=1) -> int:
    n, m = len(q), len(t)
    if n == 0 or m == 0:
        return 0
    # V = max(M, Ix, Iy) with the free borders; only two rows are kept
    Vp = [0] * (m + 1); Ixp = [NEG] * (m + 1)
    best_last_col = NEG
    for i in range(1, n + 1):
        V = [0] * (m + 1); Ix = [NEG] * (m + 1)
        iy = NEG
        qi = q[i - 1]
        for j in range(1, m + 1):
            mm = Vp[j - 1] + sub_score(qi, t[j - 1], match, mismatch)
            ix = max(Vp[j] - gap_open, Ixp[j] - gap_ext)          # gap column under q[i-1]: opened from any state of (i-1, j) or extended
            iy = max(V[j - 1] - gap_open, iy - gap_ext)            # gap column under t[j-1]
            Ix[j] = ix
            V[j] = max(mm, ix, iy)
        best_last_col = max(best_last_col, V[m])
        Vp, Ixp = V, Ix
    return max(max(Vp[1:]), best_last_col)


def cigar_columns(ops: str):
    """run-length free op string ('=XID' per column) -> validated list"""
    assert set(ops) <= set("=XID"), ops
    return list(ops)


def score_of_alignment(q: str, t: str, ops: str, match=2, mismatch=-2, gap_open=3, gap_ext=1) -> int:
    """Score of the alignment described by one op per column ('=' / 'X' consume both, 'I' consumes q, 'D' consumes t), with the
    leading and trailing gap runs free (semi-global).  Also checks that the ops consume both sequences exactly and that '=' / 'X'
    agree with the raw characters."""
    cols = cigar_columns(ops)
    # free end gaps: ONE leading run and ONE trailing run of a single gap op (a path starts on the top row or the left column and ends on
    # the bottom row or the right column; a second, different gap run next to it is an interior gap and is paid for)
    a = 0
    if cols and cols[0] in "ID":
        while a < len(cols) and cols[a] == cols[0]:
            a += 1
    b = len(cols)
    if b > a and cols[b - 1] in "ID":
        last = cols[b - 1]
        while b > a and cols[b - 1] == last:
            b -= 1
    i = sum(1 for c in cols[:a] if c == "I"); j = sum(1 for c in cols[:a] if c == "D")
    s = 0; prev = None
    for c in cols[a:b]:
        if c in "=X":
            assert (q[i] == t[j]) == (c == "="), "op %r at q[%d]=%r t[%d]=%r" % (c, i, q[i], j, t[j])
            s += sub_score(q[i], t[j], match, mismatch); i += 1; j += 1
        elif c == "I":
            s -= gap_ext if prev == "I" else gap_open; i += 1
        else:
            s -= gap_ext if prev == "D" else gap_open; j += 1
        prev = c
    i += sum(1 for c in cols[b:] if c == "I"); j += sum(1 for c in cols[b:] if c == "D")
    assert i == len(q) and j == len(t), "ops consume %d / %d of %d / %d" % (i, j, len(q), len(t))
    return s
