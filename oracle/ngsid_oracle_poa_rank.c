/* ngsid_oracle_poa_rank.c - the POA tile engine of ngsid_oracle_poa.c restated on a RANK-ORDERED graph.
 * TEST INFRASTRUCTURE ONLY (see ngsid_oracle.h).  PARITY UNPINNED like the rest of the consensus half (ngsid_oracle_poa.c header).
 *
 * ngsid_oracle_poa.c DEFINES the semantics on a node-indexed graph (node ids in creation order, edge lists, order[] / rank[]).
 * This file computes the same thing on the representation csrc/k_poa.hip uses since round 4: every per-node array is indexed by the
 * node's TOPOLOGICAL RANK, there are no node ids and no edge objects:
 *   code[r], anchor[r], cov[r]            letter, coordinate in the first sequence, count weight
 *   p0[r], p1[r], w0[r], w1[r]            ranks of the tails of the first two in-edges IN CREATION ORDER (-1 = none) and their weights
 *   many[r] + overflow list               third and later in-edges (head, tail, weight), in creation order
 *   ring[r]                               rank of the next node aligned to the same column (itself when alone)
 *   hasout[r]                             the node has an out-edge (not a sink)
 *   far[r]                                some successor lies more than HR ranks behind (kernel: keep an HBM copy of the DP row)
 * Merging an alignment inserts the new nodes at their ranks and moves every later record up by shift[r] = number of new nodes
 * inserted at or before old rank r; rank-valued fields (p0, p1, ring, overflow entries) are remapped with the same table.  The DP,
 * the tie-breaks (in-edges in creation order), the sibling rule and the heaviest bundle are those of ngsid_oracle_poa.c, line by line;
 * tests/test_consensus_oracle.py compares the two engines byte for byte.  far[] has no influence on any result; the model keeps it
 * to check the invariant the kernel relies on (a predecessor further than HR ranks back always carries the flag).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "ngsid_oracle_poa_int.h"

#define HR 8      /* csrc/k_poa.hip: DP rows kept in the LDS ring */

typedef struct {
    int V, E, capV, L0; uint64_t cw_sum;
    uint8_t* code; int* anchor; int* p0; int* p1; int64_t* w0; int64_t* w1; int* ring; uint32_t* cov; uint8_t* hasout; uint8_t* many; uint8_t* far;
    int nov, capov; int* ov_head; int* ov_tail; int64_t* ov_w;
} rgraph;

static void rg_init(rgraph* G, int capV) {
    memset(G, 0, sizeof *G); G->capV = capV; const size_t n = (size_t)capV + 1;
    G->code = malloc(n); G->anchor = malloc(sizeof(int) * n); G->p0 = malloc(sizeof(int) * n); G->p1 = malloc(sizeof(int) * n);
    G->w0 = malloc(sizeof(int64_t) * n); G->w1 = malloc(sizeof(int64_t) * n); G->ring = malloc(sizeof(int) * n); G->cov = malloc(sizeof(uint32_t) * n);
    G->hasout = malloc(n); G->many = malloc(n); G->far = malloc(n);
    G->capov = 64; G->ov_head = malloc(sizeof(int) * 64); G->ov_tail = malloc(sizeof(int) * 64); G->ov_w = malloc(sizeof(int64_t) * 64);
}
static void rg_free(rgraph* G) {
    free(G->code); free(G->anchor); free(G->p0); free(G->p1); free(G->w0); free(G->w1); free(G->ring); free(G->cov); free(G->hasout); free(G->many); free(G->far);
    free(G->ov_head); free(G->ov_tail); free(G->ov_w);
}
static void rg_reset(rgraph* G) { G->V = 0; G->E = 0; G->L0 = 0; G->cw_sum = 0; G->nov = 0; }

/* in-edge `slot` of rank r in creation order: 0 / 1 = the inline pair, 2.. = the overflow entries of r in list order */
static int rg_pred(const rgraph* G, int r, int slot, int* pr, int64_t* w) {
    if (slot == 0) { if (G->p0[r] < 0) return 0; *pr = G->p0[r]; *w = G->w0[r]; return 1; }
    if (slot == 1) { if (G->p1[r] < 0) return 0; *pr = G->p1[r]; *w = G->w1[r]; return 1; }
    if (!G->many[r]) return 0;
    int k = slot - 2;
    for (int x = 0; x < G->nov; ++x) if (G->ov_head[x] == r) { if (k == 0) { *pr = G->ov_tail[x]; *w = G->ov_w[x]; return 1; } --k; }
    return 0;
}

static void rg_add_first(rgraph* G, const pseq* S) {
    for (int i = 0; i < S->len; ++i) {
        G->code[i] = S->s[i]; G->anchor[i] = i; G->ring[i] = i; G->cov[i] = S->cw; G->many[i] = 0; G->far[i] = 0;
        G->p0[i] = i ? i - 1 : -1; G->w0[i] = i ? (int64_t)wt(S, i - 1) + wt(S, i) : 0; G->p1[i] = -1; G->w1[i] = 0; G->hasout[i] = i + 1 < S->len;
    }
    G->V = S->len; G->E = S->len > 0 ? S->len - 1 : 0; G->L0 = S->len; G->cw_sum += S->cw; G->nov = 0;
}

static inline int rg_band_lo(const rgraph* G, const pseq* S, int r, int BW) {
    int a0 = S->a0, a1 = S->a1; if (a1 < a0) { a0 = 0; a1 = G->L0 - 1; }
    long span = (long)a1 - a0 + 1; if (span < 1) span = 1;
    long c = ((long)(G->anchor[r] - a0) * (long)S->len) / span;
    long lo = c - BW / 2; long mx = (long)S->len + 1 - BW; if (mx < 0) mx = 0;
    if (lo < 0) lo = 0;
    if (lo > mx) lo = mx;
    return (int)lo;
}

/* ngsid_oracle_poa.c: poa_align with ranks in place of nodes (path[].node = rank) */
static int rg_align(const rgraph* G, const pseq* S, int m, int n, int g, int BW, ppair* path, int* npath, int* edge) {
    const int V = G->V, L = S->len, mode = S->mode;
    int* H = malloc(sizeof(int) * (size_t)V * (size_t)BW); uint8_t* dir = malloc((size_t)V * (size_t)BW); int* lo = malloc(sizeof(int) * (size_t)V);
    for (int r = 0; r < V; ++r) lo[r] = rg_band_lo(G, S, r, BW);
    int best = PNEG, br = -1, bc = -1;
    for (int r = 0; r < V; ++r) {
        const int l0 = lo[r]; int* Hr = H + (size_t)r * BW; uint8_t* Dr = dir + (size_t)r * BW;
        const int nopred = G->p0[r] < 0;
        const int use_src = nopred || mode == NGSID_POA_SEMI;
        { int pr; int64_t w_; for (int slot = 0; rg_pred(G, r, slot, &pr, &w_); ++slot) {
            if (pr >= r) { fprintf(stderr, "ngsid oracle (rank engine): topological order violated (pred rank %d >= %d)\n", pr, r); abort(); }
            if (r - pr > HR && !G->far[pr]) { fprintf(stderr, "ngsid oracle (rank engine): far flag missing on rank %d (successor %d)\n", pr, r); abort(); } } }
        for (int c = 0; c < BW; ++c) {
            const int j = l0 + c;
            if (j > L) { Hr[c] = PNEG; Dr[c] = 3; continue; }
            int bestv = PNEG, bd = 3, pr; int64_t w_;
            if (j >= 1) {
                const int sc = (G->code[r] == S->s[j - 1]) ? m : n;
                for (int slot = 0; rg_pred(G, r, slot, &pr, &w_); ++slot) {
                    const int pc = j - 1 - lo[pr];
                    if (pc < 0 || pc >= BW) continue;
                    const int hv = H[(size_t)pr * BW + pc]; if (hv <= PNEG) continue;
                    if (hv + sc > bestv) { bestv = hv + sc; bd = 0 | (slot << 2); }
                }
                if (use_src) { const int sv = (mode == NGSID_POA_LOCAL) ? 0 : (j - 1) * g; if (sv + sc > bestv) { bestv = sv + sc; bd = 0 | (SRC_SLOT << 2); } }
            }
            for (int slot = 0; rg_pred(G, r, slot, &pr, &w_); ++slot) {
                const int pc = j - lo[pr];
                if (pc < 0 || pc >= BW) continue;
                const int hv = H[(size_t)pr * BW + pc]; if (hv <= PNEG) continue;
                if (hv + g > bestv) { bestv = hv + g; bd = 1 | (slot << 2); }
            }
            if (nopred && mode != NGSID_POA_SEMI) { const int sv = (mode == NGSID_POA_LOCAL) ? 0 : j * g; if (sv + g > bestv) { bestv = sv + g; bd = 1 | (SRC_SLOT << 2); } }
            if (c >= 1 && Hr[c - 1] > PNEG && Hr[c - 1] + g > bestv) { bestv = Hr[c - 1] + g; bd = 2; }
            if (mode == NGSID_POA_LOCAL && bestv <= 0) { bestv = 0; bd = 3; }
            Hr[c] = bestv; Dr[c] = (uint8_t)bd;
            if (bestv > PNEG) {
                if (mode == NGSID_POA_LOCAL) { if (bestv > best) { best = bestv; br = r; bc = c; } }
                else if (j == L && (mode == NGSID_POA_SEMI || !G->hasout[r])) { if (bestv > best) { best = bestv; br = r; bc = c; } }
            }
        }
    }
    int np = 0, ok = 1;
    if (br < 0 || (mode == NGSID_POA_LOCAL && best <= 0)) {
        if (mode == NGSID_POA_LOCAL) { for (int i = 0; i < L; ++i) { path[np].node = -1; path[np].pos = i; ++np; } }
        else ok = 0;
    } else {
        int r = br, c = bc, jend = lo[br] + bc;
        ppair* rev = malloc(sizeof(ppair) * (size_t)(L + V + 2)); int nr = 0;
        for (int i = L - 1; i >= jend; --i) { rev[nr].node = -1; rev[nr].pos = i; ++nr; }
        int j = jend;
        for (;;) {
            const int d = dir[(size_t)r * BW + c]; const int type = d & 3, slot = d >> 2;
            if ((c == 0 && lo[r] > 0) || (c == BW - 1 && lo[r] + BW - 1 < L)) *edge |= 1;
            if (type == 3) break;
            if (type == 2) { rev[nr].node = -1; rev[nr].pos = j - 1; ++nr; --j; --c; continue; }
            if (type == 0) { rev[nr].node = r; rev[nr].pos = j - 1; ++nr; --j; } else { rev[nr].node = r; rev[nr].pos = -1; ++nr; }
            if (slot == SRC_SLOT) break;
            int pr; int64_t w_; if (!rg_pred(G, r, slot, &pr, &w_)) { fprintf(stderr, "ngsid oracle (rank engine): traceback slot %d of rank %d does not exist\n", slot, r); abort(); }
            r = pr; c = j - lo[r];
        }
        for (int i = j - 1; i >= 0; --i) { rev[nr].node = -1; rev[nr].pos = i; ++nr; }
        for (int i = nr - 1; i >= 0; --i) path[np++] = rev[i];
        free(rev);
    }
    *npath = np;
    free(H); free(dir); free(lo);
    return ok;
}

/* edge a -> b (final ranks) of weight w: the in-list of b is searched for the tail a (the node engine searches the out-list of a for the head b: same edge) */
static void rg_add_edge(rgraph* G, int a, int b, int64_t w) {
    if (G->p0[b] == a) { G->w0[b] += w; return; }
    if (G->p1[b] == a) { G->w1[b] += w; return; }
    if (G->many[b]) for (int x = 0; x < G->nov; ++x) if (G->ov_head[x] == b && G->ov_tail[x] == a) { G->ov_w[x] += w; return; }
    if (G->p0[b] < 0) { G->p0[b] = a; G->w0[b] = w; }
    else if (G->p1[b] < 0) { G->p1[b] = a; G->w1[b] = w; }
    else {
        if (G->nov == G->capov) { G->capov *= 2; G->ov_head = realloc(G->ov_head, sizeof(int) * (size_t)G->capov); G->ov_tail = realloc(G->ov_tail, sizeof(int) * (size_t)G->capov); G->ov_w = realloc(G->ov_w, sizeof(int64_t) * (size_t)G->capov); }
        G->ov_head[G->nov] = b; G->ov_tail[G->nov] = a; G->ov_w[G->nov] = w; G->nov++; G->many[b] = 1;
    }
    G->E++; G->hasout[a] = 1; if (b - a > HR) G->far[a] = 1;
}

/* ngsid_oracle_poa.c: g_add_alignment.  path[].node are ranks of the graph before the merge. */
static int rg_add_alignment(rgraph* G, const pseq* S, const ppair* path, int np) {
    const int L = S->len, V0 = G->V;
    int* al = malloc(sizeof(int) * (size_t)(L + 1)); int* nf = malloc(sizeof(int) * (size_t)(L + 1)); int* ref = malloc(sizeof(int) * (size_t)(L + 1)); int* fin = malloc(sizeof(int) * (size_t)(L + 1));
    for (int i = 0; i < L; ++i) al[i] = -1;
    for (int p = 0; p < np; ++p) if (path[p].pos >= 0 && path[p].node >= 0) al[path[p].pos] = path[p].node;
    /* A: the existing node per position (the aligned node, or a sibling whose rank lies strictly between the previous aligned position's node and this one) */
    int nnew = 0, prev_rank = -1;
    for (int i = 0; i < L; ++i) {
        int v = al[i], found = -1;
        if (v >= 0) {
            if (G->code[v] == S->s[i]) found = v;
            else for (int u = G->ring[v]; u != v; u = G->ring[u]) if (G->code[u] == S->s[i] && u > prev_rank && u < v) { found = u; break; }
            prev_rank = v;
        }
        nf[i] = found; if (found < 0) ++nnew;
    }
    if (V0 + nnew > G->capV || G->E + L > 3 * G->capV / 2) { free(al); free(nf); free(ref); free(fin); return 0; }
    { int nx = -1; for (int i = L - 1; i >= 0; --i) { if (al[i] >= 0) nx = nf[i] >= 0 ? nf[i] : al[i]; ref[i] = nx; } }
    /* C: new nodes in sequence order: the k-th goes immediately before old rank ins[k] (V0 = the end) and lands on rank ins[k] + k */
    int* ins = malloc(sizeof(int) * (size_t)(nnew + 1)); int* nanc = malloc(sizeof(int) * (size_t)(nnew + 1)); int* npos = malloc(sizeof(int) * (size_t)(nnew + 1)); int nn = 0, lastal = -1;
    for (int i = 0; i < L; ++i) {
        if (al[i] >= 0) lastal = al[i];
        if (nf[i] >= 0) continue;
        nanc[nn] = lastal >= 0 ? G->anchor[lastal] : (ref[i] >= 0 ? G->anchor[ref[i]] : (S->a1 < S->a0 ? 0 : S->a0));
        ins[nn] = ref[i] >= 0 ? ref[i] : V0; npos[nn] = i; ++nn;
    }
    /* S: shift[r] = new nodes inserted at or before old rank r.  ins[] is non-decreasing, so the LAST node of every run of equal ins writes its count
       (k + 1) and a running maximum fills the gaps (kernel: one u16 store per run, one max-scan; no atomics) */
    int* shift = calloc((size_t)V0 + 1, sizeof(int));
    for (int k = 0; k < nn; ++k) { if (k + 1 == nn || ins[k + 1] != ins[k]) shift[ins[k]] = k + 1; if (k && ins[k] < ins[k - 1]) { fprintf(stderr, "ngsid oracle (rank engine): insertion points not sorted\n"); abort(); } }
    for (int r = 1; r <= V0; ++r) if (shift[r] < shift[r - 1]) shift[r] = shift[r - 1];
#define RM(x) ((x) + shift[x])
    if (nn) {
        /* far[] is recomputed by the move pass (it visits every edge) and by the new edges of E */
        for (int r = 0; r < V0 + nn; ++r) G->far[r] = 0;
        /* D: old records move up, highest rank first (in place); rank-valued fields are remapped */
        for (int r = V0 - 1; r >= 0; --r) {
            const int nr = RM(r); const int a = G->p0[r], b = G->p1[r], rg = G->ring[r];
            G->code[nr] = G->code[r]; G->anchor[nr] = G->anchor[r]; G->cov[nr] = G->cov[r]; G->hasout[nr] = G->hasout[r]; G->many[nr] = G->many[r]; G->w0[nr] = G->w0[r]; G->w1[nr] = G->w1[r];
            G->p0[nr] = a < 0 ? -1 : RM(a); G->p1[nr] = b < 0 ? -1 : RM(b); G->ring[nr] = RM(rg);
            if (a >= 0 && nr - RM(a) > HR) G->far[RM(a)] = 1;
            if (b >= 0 && nr - RM(b) > HR) G->far[RM(b)] = 1;
        }
        for (int x = 0; x < G->nov; ++x) { G->ov_head[x] = RM(G->ov_head[x]); G->ov_tail[x] = RM(G->ov_tail[x]); if (G->ov_head[x] - G->ov_tail[x] > HR) G->far[G->ov_tail[x]] = 1; }
    }
    /* N: records of the new nodes; a new node aligned to v joins v's ring right behind v */
    for (int k = 0; k < nn; ++k) {
        const int y = ins[k] + k, i = npos[k];
        G->code[y] = S->s[i]; G->anchor[y] = nanc[k]; G->p0[y] = G->p1[y] = -1; G->w0[y] = G->w1[y] = 0; G->cov[y] = 0; G->hasout[y] = 0; G->many[y] = 0; G->ring[y] = y;
        if (al[i] >= 0) { const int v = RM(al[i]); G->ring[y] = G->ring[v]; G->ring[v] = y; }
        fin[i] = y;
    }
    for (int i = 0; i < L; ++i) if (nf[i] >= 0) fin[i] = RM(nf[i]);
#undef RM
    G->V = V0 + nn;
    /* E: coverage and edges along the sequence (final ranks) */
    for (int i = 0; i < L; ++i) { G->cov[fin[i]] += S->cw; if (i) rg_add_edge(G, fin[i - 1], fin[i], (int64_t)wt(S, i - 1) + wt(S, i)); }
    G->cw_sum += S->cw;
    free(al); free(nf); free(ref); free(fin); free(ins); free(nanc); free(npos); free(shift);
    return 1;
}

/* ngsid_oracle_poa.c: g_consensus in rank space */
static int rg_consensus(const rgraph* G, uint8_t* out, uint32_t* cov_out, int* anc_out) {
    const int V = G->V; if (V == 0) return 0;
    int* pred = malloc(sizeof(int) * (size_t)V); int64_t* sc = malloc(sizeof(int64_t) * (size_t)V);
    for (int r = 0; r < V; ++r) { pred[r] = -1; sc[r] = -1; }
    int mx = -1, t; int64_t w;
    for (int r = 0; r < V; ++r) {
        for (int slot = 0; rg_pred(G, r, slot, &t, &w); ++slot)
            if (sc[r] < w || (sc[r] == w && sc[pred[r]] <= sc[t])) { sc[r] = w; pred[r] = t; }
        if (pred[r] != -1) sc[r] += sc[pred[r]];
        if (mx < 0 || sc[mx] < sc[r]) mx = r;
    }
    while (G->hasout[mx]) {            /* branch completion */
        const int start = mx;
        for (int h = start + 1; h < V; ++h) {
            int is_succ = 0;
            for (int slot = 0; rg_pred(G, h, slot, &t, &w); ++slot) if (t == start) is_succ = 1;
            if (is_succ) for (int slot = 0; rg_pred(G, h, slot, &t, &w); ++slot) if (t != start) sc[t] = -1;
        }
        int m2 = -1;
        for (int r = start + 1; r < V; ++r) {
            sc[r] = -1; pred[r] = -1;
            for (int slot = 0; rg_pred(G, r, slot, &t, &w); ++slot) {
                if (sc[t] == -1) continue;
                if (sc[r] < w || (sc[r] == w && sc[pred[r]] <= sc[t])) { sc[r] = w; pred[r] = t; }
            }
            if (pred[r] != -1) sc[r] += sc[pred[r]];
            if (m2 < 0 || sc[m2] < sc[r]) m2 = r;
        }
        if (m2 < 0) break;
        mx = m2;
    }
    int n = 0; for (int r = mx; r != -1; r = pred[r]) ++n;
    int i = n; for (int r = mx; r != -1; r = pred[r]) { --i; out[i] = G->code[r]; if (anc_out) anc_out[i] = G->anchor[r]; if (cov_out) { uint32_t c = G->cov[r]; for (int u = G->ring[r]; u != r; u = G->ring[u]) c += G->cov[u]; cov_out[i] = c; } }
    free(pred); free(sc);
    return n;
}

/* ngsid_oracle_poa.c: the EMIT step of run_tile_band (end trim by coverage, one-third rule of the upper levels, span of the output) */
static void rg_emit(const rgraph* G, int members, const pprm* P, int want_cov, pout* outs, int* nout) {
    if (!(G->V > 0 && members > 0)) return;
    pout* o = &outs[(*nout)++]; o->s = malloc((size_t)G->V + 1); o->cov = (want_cov || P->trim_tiles) ? malloc(sizeof(uint32_t) * ((size_t)G->V + 1)) : NULL; int* anc_ = malloc(sizeof(int) * ((size_t)G->V + 1));
    o->len = rg_consensus(G, o->s, o->cov, anc_); o->cw = G->cw_sum; int b_ = 0, e_ = o->len - 1;
    if (P->trim_tiles && o->len > 0) {
        uint32_t thr = (uint32_t)(G->cw_sum / 2); for (; b_ < o->len; ++b_) if (o->cov[b_] >= thr) break; for (; e_ >= 0; --e_) if (o->cov[e_] >= thr) break;
        if (b_ < e_) {
            const uint32_t thr3 = (P->trim_tiles & 2) ? (uint32_t)(G->cw_sum / 3) : 0u;
            int k_ = 0; for (int x_ = b_; x_ <= e_; ++x_) if (o->cov[x_] >= thr3) { o->s[k_] = o->s[x_]; o->cov[k_] = o->cov[x_]; ++k_; } o->len = k_; } else { b_ = 0; e_ = o->len - 1; } }
    o->a0 = o->len > 0 ? anc_[b_] : 0; o->a1 = o->len > 0 ? anc_[e_] : -1; free(anc_);
    if (!want_cov) { free(o->cov); o->cov = NULL; }
}

int run_tile_band_rank(const pseq* seqs, int ns, const pseq* backbone, const pprm* P, int band, int* edge, pout* outs, int want_cov) {
    int nout = 0, maxlen = backbone ? backbone->len : 0;
    for (int i = 0; i < ns; ++i) if (seqs[i].len > maxlen) maxlen = seqs[i].len;
    int L0 = backbone ? backbone->len : (ns ? seqs[0].len : 0);
    int capV = cap_for(L0 > 0 ? L0 : 1, P->node_cap);
    rgraph G; rg_init(&G, capV > maxlen + 1 ? capV : maxlen + 1);
    ppair* path = malloc(sizeof(ppair) * (size_t)(maxlen + G.capV + 4));
    int members = 0;
    for (int i = 0; i < ns; ++i) {
        const pseq* S = &seqs[i];
        if (S->len <= 0) continue;
        if (G.V == 0) {
            if (backbone) { rg_add_first(&G, backbone); }
            else { if (S->len > G.capV) continue; rg_add_first(&G, S); members = 1; continue; }
        }
        int np = 0;
        int ok = rg_align(&G, S, P->m, P->n, P->g, band, path, &np, edge);
        if (!ok) continue;
        if (!rg_add_alignment(&G, S, path, np)) {
            rg_emit(&G, members, P, want_cov, outs, &nout); rg_reset(&G); members = 0;
            if (backbone) { rg_add_first(&G, backbone); ok = rg_align(&G, S, P->m, P->n, P->g, band, path, &np, edge); if (ok && rg_add_alignment(&G, S, path, np)) members = 1; }
            else if (S->len <= G.capV) { rg_add_first(&G, S); members = 1; }
            continue;
        }
        members++;
    }
    rg_emit(&G, members, P, want_cov, outs, &nout);
    free(path); rg_free(&G);
    return nout;
}
