/* ngsid_oracle_poa.c - CPU restatement of the consensus + polishing path (spoa-style POA, racon-style windows).
 * TEST INFRASTRUCTURE ONLY (see ngsid_oracle.h).
 *
 * PARITY UNPINNED: spoa 4.0.7, racon 1.4.20 and minimap2 2.23 are external binaries that are not in
 * /root/reference and not in this image (call sites consensus.py:87,121,122; the reference holds no test
 * vectors for them).  What follows restates their published algorithms [from memory of the upstream
 * sources] with this build's documented choices, and is pinned by synthetic ground truth
 * (tests/test_consensus_*.py: consensus == generating amplicon).
 *
 *   spoa  `spoa reads.fq -l 0 -r 0 -g -2`: m=+5 n=-4 g=-2, e=-6 => g>=e selects LINEAR gaps; local (SW)
 *         alignment of each read to the DAG in file order; FASTQ => edge weight += (q[i-1]-33)+(q[i]-33);
 *         mismatching base -> aligned sibling node; unaligned head/tail -> new branch;
 *         consensus = heaviest bundle with branch completion.
 *   racon defaults -w 500 -q 10 -e 0.3 -m 3 -x -5 -g -4: windows of the backbone, backbone first with
 *         zero weight, layers added by global alignment (full-span) or to the spanned sub-graph
 *         (here: graph ends free), consensus = heaviest bundle; TGS windows are end-trimmed by coverage.
 *
 * BUILD CHOICES (shared, bit for bit, with the HIP kernels in ngspeciesid_amd/csrc/k_poa.hip):
 *   - banded DP: row of node v covers `band` columns centred on the node's anchor (its coordinate in the
 *     first sequence of the graph) scaled to the aligned sequence;
 *   - topological order is maintained incrementally: a new node is inserted immediately before the next
 *     already-existing node the alignment uses (or at the end);
 *   - depth tiling: a group is consumed in tiles of `tile_depth` sequences in file order; every tile is an
 *     exact-order POA; tile consensuses (weighted by the reads they stand for) are merged by the same
 *     procedure, level by level.  tile_depth<=0 means one tile = plain spoa order for the whole group;
 *   - a graph that would exceed its node capacity is closed (its consensus is emitted) and a new graph is
 *     started with the sequence that did not fit;
 *   - (round 3) a level's last tile takes a remainder of fewer than (D + 1) / 2 sequences along (ntiles_of), and tile
 *     consensuses of the UPPER levels drop interior bases whose column carries less than a third of the merged weight
 *     (EMIT, trim_tiles & 2): the heaviest bundle maximises the SUM of edge weights, so in a tile of two or three heavy
 *     members a minority's k-base insertion wins once (k + 1) w > W - it would never in one graph of all reads.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include "ngsid_oracle.h"

#include "ngsid_oracle_poa_int.h"

/* ---------------------------------------------------------------- graph */
typedef struct {
    int V, E, capV, capE;
    uint8_t* code; int* anchor; int* in_first; int* in_last; int* out_first; int* out_last; int* ring; uint32_t* cov;
    int* e_tail; int* e_head; int64_t* e_w; int* e_next_in; int* e_next_out;
    int* order; int* rank;
    int L0;            /* length of the first sequence (anchor coordinate range) */
    uint64_t cw_sum;   /* count weight of the sequences merged (backbone excluded) */
} graph;

static void g_init(graph* G, int capV) {
    memset(G, 0, sizeof *G); G->capV = capV; G->capE = capV * 4 + 16;
    G->code = malloc((size_t)capV); G->anchor = malloc(sizeof(int) * (size_t)capV); G->in_first = malloc(sizeof(int) * (size_t)capV); G->in_last = malloc(sizeof(int) * (size_t)capV);
    G->out_first = malloc(sizeof(int) * (size_t)capV); G->out_last = malloc(sizeof(int) * (size_t)capV); G->ring = malloc(sizeof(int) * (size_t)capV); G->cov = malloc(sizeof(uint32_t) * (size_t)capV);
    G->order = malloc(sizeof(int) * (size_t)capV); G->rank = malloc(sizeof(int) * (size_t)capV);
    G->e_tail = malloc(sizeof(int) * (size_t)G->capE); G->e_head = malloc(sizeof(int) * (size_t)G->capE); G->e_w = malloc(sizeof(int64_t) * (size_t)G->capE);
    G->e_next_in = malloc(sizeof(int) * (size_t)G->capE); G->e_next_out = malloc(sizeof(int) * (size_t)G->capE);
}
static void g_free(graph* G) {
    free(G->code); free(G->anchor); free(G->in_first); free(G->in_last); free(G->out_first); free(G->out_last); free(G->ring); free(G->cov);
    free(G->order); free(G->rank); free(G->e_tail); free(G->e_head); free(G->e_w); free(G->e_next_in); free(G->e_next_out);
}
static void g_reset(graph* G) { G->V = 0; G->E = 0; G->L0 = 0; G->cw_sum = 0; }
static int g_new_node(graph* G, uint8_t c, int anchor) {
    int v = G->V++; G->code[v] = c; G->anchor[v] = anchor; G->in_first[v] = G->in_last[v] = G->out_first[v] = G->out_last[v] = -1; G->ring[v] = v; G->cov[v] = 0; return v;
}
static void g_add_edge(graph* G, int a, int b, int64_t w) {
    for (int e = G->out_first[a]; e >= 0; e = G->e_next_out[e]) if (G->e_head[e] == b) { G->e_w[e] += w; return; }
    if (G->E == G->capE) { G->capE *= 2; G->e_tail = realloc(G->e_tail, sizeof(int) * (size_t)G->capE); G->e_head = realloc(G->e_head, sizeof(int) * (size_t)G->capE); G->e_w = realloc(G->e_w, sizeof(int64_t) * (size_t)G->capE);
        G->e_next_in = realloc(G->e_next_in, sizeof(int) * (size_t)G->capE); G->e_next_out = realloc(G->e_next_out, sizeof(int) * (size_t)G->capE); }
    int e = G->E++; G->e_tail[e] = a; G->e_head[e] = b; G->e_w[e] = w; G->e_next_in[e] = -1; G->e_next_out[e] = -1;
    if (G->out_last[a] < 0) G->out_first[a] = e; else G->e_next_out[G->out_last[a]] = e; G->out_last[a] = e;
    if (G->in_last[b] < 0) G->in_first[b] = e; else G->e_next_in[G->in_last[b]] = e; G->in_last[b] = e;
}


static void g_add_first(graph* G, const pseq* S) {
    for (int i = 0; i < S->len; ++i) { int v = g_new_node(G, S->s[i], i); G->order[v] = v; G->rank[v] = v; G->cov[v] = S->cw; if (i) g_add_edge(G, v - 1, v, (int64_t)wt(S, i - 1) + wt(S, i)); }
    G->L0 = S->len; G->cw_sum += S->cw;
}

/* ---------------------------------------------------------------- banded alignment of one sequence to the graph
 * H[r][c]: r = rank, c = column - lo[r]; columns 0..L (column j = j bases consumed).  Returns path pairs in forward order. */

static inline int band_lo(const graph* G, const pseq* S, int v, int BW) {
    int a0 = S->a0, a1 = S->a1; if (a1 < a0) { a0 = 0; a1 = G->L0 - 1; }
    long span = (long)a1 - a0 + 1; if (span < 1) span = 1;
    long c = ((long)(G->anchor[v] - a0) * (long)S->len) / span;
    long lo = c - BW / 2; long mx = (long)S->len + 1 - BW; if (mx < 0) mx = 0;
    if (lo < 0) lo = 0; if (lo > mx) lo = mx;
    return (int)lo;
}

/* POA_SUBGRAPH (oracle only, ongsid_debug_polish_rules bit 1): racon's alignment of a layer that does not span its window - global (NW) alignment to the SUB-graph
   between the backbone positions [a0, a1] of the layer (racon src/window.cpp: graph->subgraph(positions.first, positions.second); spoa Graph::subgraph /
   extract_subgraph_nodes: every node reached backwards - in-edges and aligned siblings - from backbone node a1, backbone nodes below a0 left out; the first sequence of
   a window graph is the backbone, so backbone position == node id).  Nodes whose predecessors all lie outside become sources, nodes without a successor inside sinks. */
#define POA_SUBGRAPH 3
static uint8_t* subgraph_mask(const graph* G, int a0, int a1) {
    uint8_t* in = calloc((size_t)G->V + 1, 1); int* st = malloc(sizeof(int) * ((size_t)G->V * 4 + (size_t)G->E + 16)); int ns = 0;
    if (a1 < 0 || a1 >= G->L0) a1 = G->L0 - 1; if (a0 < 0) a0 = 0;
    st[ns++] = a1;
    while (ns) { const int v = st[--ns]; if (in[v] || v < a0) continue; in[v] = 1;
        for (int e = G->in_first[v]; e >= 0; e = G->e_next_in[e]) if (!in[G->e_tail[e]]) st[ns++] = G->e_tail[e];
        for (int u = G->ring[v]; u != v; u = G->ring[u]) if (!in[u]) st[ns++] = u; }
    free(st); return in;
}
static int poa_align(const graph* G, const pseq* S, int m, int n, int g, int BW, ppair* path /* cap len+V */, int* npath, int* edge /* |= 1 when the traceback visits a clipped band-edge cell */) {
    const int V = G->V, L = S->len, sub = S->mode == POA_SUBGRAPH, mode = sub ? NGSID_POA_GLOBAL : S->mode;
    uint8_t* insub = sub ? subgraph_mask(G, S->a0, S->a1) : NULL;
    int* H = malloc(sizeof(int) * (size_t)V * (size_t)BW); uint8_t* dir = malloc((size_t)V * (size_t)BW); int* lo = malloc(sizeof(int) * (size_t)V);
    for (int r = 0; r < V; ++r) lo[r] = band_lo(G, S, G->order[r], BW);
    int best = PNEG, br = -1, bc = -1;
    for (int r = 0; r < V; ++r) {
        const int v = G->order[r]; const int l0 = lo[r]; int* Hr = H + (size_t)r * BW; uint8_t* Dr = dir + (size_t)r * BW;
        int nopred = G->in_first[v] < 0, nosucc = G->out_first[v] < 0;
        if (sub) {
            if (!insub[v]) { for (int c = 0; c < BW; ++c) { Hr[c] = PNEG; Dr[c] = 3; } continue; }
            nopred = 1; for (int e = G->in_first[v]; e >= 0; e = G->e_next_in[e]) if (insub[G->e_tail[e]]) { nopred = 0; break; }
            nosucc = 1; for (int e = G->out_first[v]; e >= 0; e = G->e_next_out[e]) if (insub[G->e_head[e]]) { nosucc = 0; break; }
        }
        const int use_src = nopred || mode == NGSID_POA_SEMI;
        for (int c = 0; c < BW; ++c) {
            const int j = l0 + c;
            if (j > L) { Hr[c] = PNEG; Dr[c] = 3; continue; }
            int bestv = PNEG, bd = 3, slot = 0;
            /* diag over predecessors in in-edge order, then the virtual source */
            if (j >= 1) {
                const int sc = (G->code[v] == S->s[j - 1]) ? m : n;
                for (int e = G->in_first[v]; e >= 0; e = G->e_next_in[e], ++slot) {
                    const int pr = G->rank[G->e_tail[e]]; const int pc = j - 1 - lo[pr];
                    if (sub && !insub[G->e_tail[e]]) continue;
                    if (pr >= r) { fprintf(stderr, "ngsid oracle: topological order violated (pred rank %d >= %d)\n", pr, r); abort(); }
                    if (pc < 0 || pc >= BW) continue;
                    const int hv = H[(size_t)pr * BW + pc]; if (hv <= PNEG) continue;
                    if (hv + sc > bestv) { bestv = hv + sc; bd = 0 | (slot << 2); }
                }
                if (use_src) { const int sv = (mode == NGSID_POA_LOCAL) ? 0 : (j - 1) * g; if (sv + sc > bestv) { bestv = sv + sc; bd = 0 | (SRC_SLOT << 2); } }
            }
            /* up (node consumed, no base) */
            slot = 0;
            for (int e = G->in_first[v]; e >= 0; e = G->e_next_in[e], ++slot) {
                const int pr = G->rank[G->e_tail[e]]; const int pc = j - lo[pr];
                if (sub && !insub[G->e_tail[e]]) continue;
                if (pc < 0 || pc >= BW) continue;
                const int hv = H[(size_t)pr * BW + pc]; if (hv <= PNEG) continue;
                if (hv + g > bestv) { bestv = hv + g; bd = 1 | (slot << 2); }
            }
            if (nopred && mode != NGSID_POA_SEMI) { const int sv = (mode == NGSID_POA_LOCAL) ? 0 : j * g; if (sv + g > bestv) { bestv = sv + g; bd = 1 | (SRC_SLOT << 2); } }
            /* left (base consumed, no node) */
            if (c >= 1 && Hr[c - 1] > PNEG && Hr[c - 1] + g > bestv) { bestv = Hr[c - 1] + g; bd = 2; }
            if (mode == NGSID_POA_LOCAL && bestv <= 0) { bestv = 0; bd = 3; }
            Hr[c] = bestv; Dr[c] = (uint8_t)bd;
            if (bestv > PNEG) {
                if (mode == NGSID_POA_LOCAL) { if (bestv > best) { best = bestv; br = r; bc = c; } }
                else if (j == L && (mode == NGSID_POA_SEMI || nosucc)) { if (bestv > best) { best = bestv; br = r; bc = c; } }
            }
        }
    }
    int np = 0, ok = 1;
    if (br < 0 || (mode == NGSID_POA_LOCAL && best <= 0)) {
        if (mode == NGSID_POA_LOCAL) { for (int i = 0; i < L; ++i) { path[np].node = -1; path[np].pos = i; ++np; } }   /* nothing aligned: whole read becomes a branch */
        else ok = 0;
    } else {
        /* walk back; collect reversed */
        int r = br, c = bc, jend = lo[br] + bc;
        ppair* rev = malloc(sizeof(ppair) * (size_t)(L + V + 2)); int nr = 0;
        for (int i = L - 1; i >= jend; --i) { rev[nr].node = -1; rev[nr].pos = i; ++nr; }        /* unaligned tail (local) */
        int j = jend;
        for (;;) {
            const int v = G->order[r]; const int d = dir[(size_t)r * BW + c]; const int type = d & 3, slot = d >> 2;
            if ((c == 0 && lo[r] > 0) || (c == BW - 1 && lo[r] + BW - 1 < L)) *edge |= 1;      /* the optimal path may continue outside the band */
            if (type == 3) break;
            if (type == 2) { rev[nr].node = -1; rev[nr].pos = j - 1; ++nr; --j; --c; continue; }
            if (type == 0) { rev[nr].node = v; rev[nr].pos = j - 1; ++nr; --j; } else { rev[nr].node = v; rev[nr].pos = -1; ++nr; }
            if (slot == SRC_SLOT) break;
            int e = G->in_first[v]; for (int t = 0; t < slot; ++t) e = G->e_next_in[e];
            r = G->rank[G->e_tail[e]]; c = j - lo[r];
        }
        for (int i = j - 1; i >= 0; --i) { rev[nr].node = -1; rev[nr].pos = i; ++nr; }           /* unaligned head / leading insertions */
        for (int i = nr - 1; i >= 0; --i) path[np++] = rev[i];
        free(rev);
    }
    *npath = np;
    free(H); free(dir); free(lo); free(insub);
    return ok;
}

/* ---------------------------------------------------------------- add an aligned sequence (returns 0 if it does not fit) */
static int g_add_alignment(graph* G, const pseq* S, const ppair* path, int np) {
    const int L = S->len, V0 = G->V;
    int* alnode = malloc(sizeof(int) * (size_t)(L + 1)); int* nodeof = malloc(sizeof(int) * (size_t)(L + 1)); int* ref = malloc(sizeof(int) * (size_t)(L + 1));
    for (int i = 0; i < L; ++i) alnode[i] = -1;
    for (int p = 0; p < np; ++p) if (path[p].pos >= 0 && path[p].node >= 0) alnode[path[p].pos] = path[p].node;
    /* existing node per position, count new */
    /* A sibling (node aligned to the same column) is reused only if its rank lies strictly between the rank of the previous aligned
       position's node and that of this position's aligned node.  With new nodes inserted immediately before the CHOSEN node of the
       next aligned position this keeps `order` a valid topological order without a re-sort (spoa re-sorts instead); a sibling outside
       that window gets a duplicate node, which only splits its weight. */
    int nnew = 0, prev_rank = -1;
    for (int i = 0; i < L; ++i) {
        int v = alnode[i], found = -1;
        if (v >= 0) {
            if (G->code[v] == S->s[i]) found = v;
            else for (int u = G->ring[v]; u != v; u = G->ring[u]) if (G->code[u] == S->s[i] && G->rank[u] > prev_rank && G->rank[u] < G->rank[v]) { found = u; break; }
            prev_rank = G->rank[v];
        }
        nodeof[i] = found; if (found < 0) ++nnew;
    }
    if (V0 + nnew > G->capV || G->E + L > 3 * G->capV / 2) { free(alnode); free(nodeof); free(ref); return 0; }   /* node / edge (1.5x) capacity */
    /* ref(i) = node chosen for the first aligned position >= i (the reused node, else the aligned node itself); new nodes are inserted
       immediately before it, -1 = end */
    { int nx = -1; for (int i = L - 1; i >= 0; --i) { if (alnode[i] >= 0) nx = nodeof[i] >= 0 ? nodeof[i] : alnode[i]; ref[i] = nx; } }
    /* create nodes in sequence order; anchor from the nearest aligned position at or before i, else after, else a0 */
    int lastal = -1;
    int* newlist = malloc(sizeof(int) * (size_t)(nnew + 1)); int* newref = malloc(sizeof(int) * (size_t)(nnew + 1)); int nn = 0;
    for (int i = 0; i < L; ++i) {
        if (alnode[i] >= 0) lastal = alnode[i];
        if (nodeof[i] >= 0) continue;
        int anc = lastal >= 0 ? G->anchor[lastal] : (ref[i] >= 0 ? G->anchor[ref[i]] : (S->a1 < S->a0 ? 0 : S->a0));
        int y = g_new_node(G, S->s[i], anc);
        if (alnode[i] >= 0) { int v = alnode[i]; G->ring[y] = G->ring[v]; G->ring[v] = y; }
        nodeof[i] = y; newlist[nn] = y; newref[nn] = ref[i]; ++nn;
    }
    /* ranks: every new node goes immediately before its ref node (in sequence order), or to the end */
    {
        int* before = calloc((size_t)V0 + 1, sizeof(int));
        for (int t = 0; t < nn; ++t) before[newref[t] >= 0 ? G->rank[newref[t]] : V0]++;
        int* norder = malloc(sizeof(int) * (size_t)G->V); int* start = malloc(sizeof(int) * ((size_t)V0 + 1)); int acc = 0;
        for (int r = 0; r <= V0; ++r) { start[r] = r + acc; acc += before[r]; if (r < V0) norder[r + acc] = G->order[r]; }
        memset(before, 0, sizeof(int) * ((size_t)V0 + 1));
        for (int t = 0; t < nn; ++t) { int r = newref[t] >= 0 ? G->rank[newref[t]] : V0; norder[start[r] + before[r]] = newlist[t]; before[r]++; }
        for (int r = 0; r < G->V; ++r) { G->order[r] = norder[r]; G->rank[norder[r]] = r; }
        free(before); free(norder); free(start);
    }
    for (int i = 0; i < L; ++i) { G->cov[nodeof[i]] += S->cw; if (i) g_add_edge(G, nodeof[i - 1], nodeof[i], (int64_t)wt(S, i - 1) + wt(S, i)); }
    G->cw_sum += S->cw;
    free(alnode); free(nodeof); free(ref); free(newlist); free(newref);
    return 1;
}

/* ---------------------------------------------------------------- heaviest bundle (spoa Graph::TraverseHeaviestBundle + BranchCompletion) */
static int g_consensus(const graph* G, uint8_t* out, uint32_t* cov_out, int* anc_out) {
    const int V = G->V; if (V == 0) return 0;
    int* pred = malloc(sizeof(int) * (size_t)V); int64_t* sc = malloc(sizeof(int64_t) * (size_t)V);
    for (int v = 0; v < V; ++v) { pred[v] = -1; sc[v] = -1; }
    int mx = -1;
    for (int r = 0; r < V; ++r) {
        int v = G->order[r];
        for (int e = G->in_first[v]; e >= 0; e = G->e_next_in[e]) {
            int t = G->e_tail[e];
            if (sc[v] < G->e_w[e] || (sc[v] == G->e_w[e] && sc[pred[v]] <= sc[t])) { sc[v] = G->e_w[e]; pred[v] = t; }
        }
        if (pred[v] != -1) sc[v] += sc[pred[v]];
        if (mx < 0 || sc[mx] < sc[v]) mx = v;
    }
    while (G->out_first[mx] >= 0) {            /* branch completion */
        int start = mx;
        for (int e = G->out_first[start]; e >= 0; e = G->e_next_out[e])
            for (int f = G->in_first[G->e_head[e]]; f >= 0; f = G->e_next_in[f]) if (G->e_tail[f] != start) sc[G->e_tail[f]] = -1;
        int m2 = -1;
        for (int r = G->rank[start] + 1; r < V; ++r) {
            int v = G->order[r]; sc[v] = -1; pred[v] = -1;
            for (int e = G->in_first[v]; e >= 0; e = G->e_next_in[e]) {
                int t = G->e_tail[e]; if (sc[t] == -1) continue;
                if (sc[v] < G->e_w[e] || (sc[v] == G->e_w[e] && sc[pred[v]] <= sc[t])) { sc[v] = G->e_w[e]; pred[v] = t; }
            }
            if (pred[v] != -1) sc[v] += sc[pred[v]];
            if (m2 < 0 || sc[m2] < sc[v]) m2 = v;
        }
        if (m2 < 0) break;
        mx = m2;
    }
    int n = 0; for (int v = mx; v != -1; v = pred[v]) ++n;
    int i = n; for (int v = mx; v != -1; v = pred[v]) { --i; out[i] = G->code[v]; if (anc_out) anc_out[i] = G->anchor[v]; if (cov_out) { uint32_t c = G->cov[v]; for (int u = G->ring[v]; u != v; u = G->ring[u]) c += G->cov[u]; cov_out[i] = c; } }
    free(pred); free(sc);
    return n;
}


/* dev aid (ODBG_CONS=1): where the consensus of a WINDOW graph leaves the backbone - the non-backbone nodes on the heaviest path, their weights and the
   weight of the backbone edge they bypass */
static void g_debug_consensus(const graph* G, int bblen) {
    const int V = G->V; int* pred = malloc(sizeof(int) * (size_t)V); int64_t* sc = malloc(sizeof(int64_t) * (size_t)V); int64_t* pw = malloc(sizeof(int64_t) * (size_t)V);
    for (int v = 0; v < V; ++v) { pred[v] = -1; sc[v] = -1; pw[v] = 0; }
    int mx = -1;
    for (int r = 0; r < V; ++r) { int v = G->order[r];
        for (int e = G->in_first[v]; e >= 0; e = G->e_next_in[e]) { int t = G->e_tail[e]; if (sc[v] < G->e_w[e] || (sc[v] == G->e_w[e] && sc[pred[v]] <= sc[t])) { sc[v] = G->e_w[e]; pred[v] = t; pw[v] = G->e_w[e]; } }
        if (pred[v] != -1) sc[v] += sc[pred[v]];
        if (mx < 0 || sc[mx] < sc[v]) mx = v; }
    fprintf(stderr, "[cons] window of %d backbone bases, %d nodes, %llu layers; best-score node %d (backbone? %d) anchor %d code %c out-edges %s\n", bblen, V, (unsigned long long)G->cw_sum, mx, mx < bblen, G->anchor[mx], G->code[mx], G->out_first[mx] >= 0 ? "yes -> branch completion" : "none");
    for (int v = mx; v != -1; v = pred[v]) if (v >= bblen) {
        fprintf(stderr, "   off-backbone node %d '%c' anchor %d cov %u ring:", v, G->code[v], G->anchor[v], G->cov[v]);
        for (int u = G->ring[v]; u != v; u = G->ring[u]) fprintf(stderr, " %d'%c'(cov %u)", u, G->code[u], G->cov[u]);
        fprintf(stderr, " | in:"); for (int e = G->in_first[v]; e >= 0; e = G->e_next_in[e]) fprintf(stderr, " %d'%c'->w%lld", G->e_tail[e], G->code[G->e_tail[e]], (long long)G->e_w[e]);
        fprintf(stderr, " | out:"); for (int e = G->out_first[v]; e >= 0; e = G->e_next_out[e]) fprintf(stderr, " ->%d'%c' w%lld", G->e_head[e], G->code[G->e_head[e]], (long long)G->e_w[e]);
        if (pred[v] >= 0) { fprintf(stderr, " | pred %d out:", pred[v]); for (int e = G->out_first[pred[v]]; e >= 0; e = G->e_next_out[e]) fprintf(stderr, " ->%d'%c' w%lld", G->e_head[e], G->code[G->e_head[e]], (long long)G->e_w[e]); }
        fprintf(stderr, "\n");
    }
    free(pred); free(sc); free(pw);
}

static int g_polish_rules = 0;      /* ongsid_debug_polish_rules, below */
/* ---------------------------------------------------------------- tile engine: sequences in order -> one or more (consensus, cw) */
/* which tile engine run_tile uses: 0 = the node-indexed graph below (the definition), 1 = the rank-ordered restatement (ngsid_oracle_poa_rank.c).
   Both give the same bytes (tests/test_consensus_oracle.py::test_rank_engine_equals_node_engine). */
static int g_poa_engine = 0;
int32_t ongsid_debug_poa_engine(int32_t e) { const int old = g_poa_engine; if (e >= 0) g_poa_engine = e; return old; }

/* returns number of outputs appended to outs (caller frees .s/.cov) */
static int run_tile_band(const pseq* seqs, int ns, const pseq* backbone, const pprm* P, int band, int* edge, pout* outs, int want_cov) {
    int nout = 0, maxlen = backbone ? backbone->len : 0;
    for (int i = 0; i < ns; ++i) if (seqs[i].len > maxlen) maxlen = seqs[i].len;
    int L0 = backbone ? backbone->len : (ns ? seqs[0].len : 0);
    int capV = cap_for(L0 > 0 ? L0 : 1, P->node_cap); if (capV < 2 * maxlen + 64) { /* capacity follows the FIRST sequence; longer members may trigger a split */ }
    graph G; g_init(&G, capV > maxlen + 1 ? capV : maxlen + 1);
    ppair* path = malloc(sizeof(ppair) * (size_t)(maxlen + G.capV + 4));
    int members = 0;
#define EMIT() do { if (G.V > 0 && members > 0) { if (backbone && getenv("ODBG_CONS")) g_debug_consensus(&G, backbone->len); pout* o = &outs[nout++]; o->s = malloc((size_t)G.V + 1); o->cov = (want_cov || P->trim_tiles) ? malloc(sizeof(uint32_t) * ((size_t)G.V + 1)) : NULL; int* anc_ = malloc(sizeof(int) * ((size_t)G.V + 1)); \
        o->len = g_consensus(&G, o->s, o->cov, anc_); o->cw = G.cw_sum; int b_ = 0, e_ = o->len - 1; \
        if (P->trim_tiles && o->len > 0) { /* coverage-trim the tile consensus ends: keeps unsupported backbone ends from propagating up the hierarchy */ \
            uint32_t thr = (uint32_t)(G.cw_sum / 2); for (; b_ < o->len; ++b_) if (o->cov[b_] >= thr) break; for (; e_ >= 0; --e_) if (o->cov[e_] >= thr) break; \
            if (b_ < e_) { /* UPPER LEVELS (trim_tiles & 2: the members are weighted tile consensuses): between the kept ends every base whose column carries less than a third of the \
                              merged weight goes as well.  The heaviest bundle maximises the SUM of the edge weights of a path: a k-base insertion carried by the weight w wins against the \
                              direct edge of the rest W as soon as (k + 1) w > W, i.e. from a third of the weight for one base (unchanged by this rule) down to a seventh for five - in ONE \
                              graph of all reads such a minority never gets near that, in a tile of two or three heavy members it does (a 5-base read tail carried by 17 % of the reads \
                              reached the top of a 37 714-layer hierarchy).  Level-0 tiles (real reads) keep spoa's / racon's behaviour. */ \
                const uint32_t thr3 = (P->trim_tiles & 2) ? (uint32_t)(G.cw_sum / 3) : 0u; \
                int k_ = 0; for (int x_ = b_; x_ <= e_; ++x_) if (o->cov[x_] >= thr3) { o->s[k_] = o->s[x_]; o->cov[k_] = o->cov[x_]; ++k_; } o->len = k_; } else { b_ = 0; e_ = o->len - 1; } } \
        o->a0 = o->len > 0 ? anc_[b_] : 0; o->a1 = o->len > 0 ? anc_[e_] : -1; free(anc_); \
        if (!want_cov) { free(o->cov); o->cov = NULL; } } } while (0)
    for (int i = 0; i < ns; ++i) {
        const pseq* S = &seqs[i];
        if (S->len <= 0) continue;
        if (G.V == 0) {
            if (backbone) { g_add_first(&G, backbone); }
            else { if (S->len > G.capV) continue; g_add_first(&G, S); members = 1; continue; }
        }
        int np = 0;
        int ok = poa_align(&G, S, P->m, P->n, P->g, band, path, &np, edge);
        if (!ok) continue;                                   /* no valid end cell inside the band: sequence dropped */
        if (backbone && getenv("ODBG_CONS") && np > 0 && (path[0].node < 0 || path[np - 1].node < 0)) { static int shown = 0; if (shown++ < 40) { fprintf(stderr, "   [layer %d] mode %d span [%d, %d] len %d head %.8s tail %.8s | path starts:", i, S->mode, S->a0, S->a1, S->len, (const char*)S->s, (const char*)S->s + (S->len > 8 ? S->len - 8 : 0)); for (int x = 0; x < 6 && x < np; ++x) fprintf(stderr, " (n%d,p%d)", path[x].node, path[x].pos); fprintf(stderr, " ends:"); for (int x = np > 4 ? np - 4 : 0; x < np; ++x) fprintf(stderr, " (n%d,p%d)", path[x].node, path[x].pos); fprintf(stderr, "\n"); } }
        pseq S2;
        if (backbone && (g_polish_rules & 4) && np > 0) {      /* PROBE (bit 2): bases of a window layer that the alignment leaves in front of the graph's first / behind its last aligned node are dropped instead of becoming new source / sink nodes */
            int h = 0, t = 0; while (h < np && path[h].node < 0) ++h; while (t < np - h && path[np - 1 - t].node < 0) ++t;
            if (h + t > 0 && h + t < S->len) { S2 = *S; S2.s += h; if (S2.q) S2.q += h; S2.len -= h + t; for (int x = h; x < np - t; ++x) { path[x - h] = path[x]; if (path[x - h].pos >= 0) path[x - h].pos -= h; } np -= h + t; S = &S2; }
        }
        if (!g_add_alignment(&G, S, path, np)) {
            /* does not fit: close this graph, start a new one with this sequence */
            EMIT(); g_reset(&G); members = 0;
            if (backbone) { g_add_first(&G, backbone); ok = poa_align(&G, S, P->m, P->n, P->g, band, path, &np, edge); if (ok && g_add_alignment(&G, S, path, np)) members = 1; }
            else if (S->len <= G.capV) { g_add_first(&G, S); members = 1; }
            continue;
        }
        members++;
    }
    EMIT();
    free(path); g_free(&G);
    return nout;
}

/* Band-edge check: a tile in which any traceback touched a clipped edge of its band is redone as a whole with twice the band (up to 256
   columns), so a path that wants to leave the band gets the room - the result then does not depend on the narrow default band. */
static int run_tile(const pseq* seqs, int ns, const pseq* backbone, const pprm* P, pout* outs, int want_cov) {
    for (int band = P->band;; band *= 2) {
        int edge = 0;
        if (getenv("ODBG_ENGINE")) { static int said[2]; if (!said[g_poa_engine != 0]) { said[g_poa_engine != 0] = 1; fprintf(stderr, "[oracle] tile engine %d in use\n", g_poa_engine); } }
        const int nout = g_poa_engine ? run_tile_band_rank(seqs, ns, backbone, P, band, &edge, outs, want_cov) : run_tile_band(seqs, ns, backbone, P, band, &edge, outs, want_cov);
        if (getenv("ODBG_EDGE")) { static long ntile = 0, nredo = 0; ++ntile; if (edge && band < 256) ++nredo; if ((ntile & 1023) == 0) fprintf(stderr, "tiles %ld redone %ld\n", ntile, nredo); }
        if (!edge || band >= 256) return nout;
        for (int i = 0; i < nout; ++i) { free(outs[i].s); free(outs[i].cov); }
    }
}

/* tiles of a level: n sequences in tiles of D in order; a remainder of fewer than (D + 1) / 2 sequences joins the last full tile instead of forming a tile of its own
   (one- and two-member remainder tiles were the weak spot of the hierarchy: one heavy minority member carries its insertions through them); tile t = [t D, t == ntiles - 1 ? n : (t + 1) D) */
static int ntiles_of(int n, int D) { if (n <= 0) return 0; if (D <= 0 || n <= D) return 1; int r = n % D; return (r != 0 && r < (D + 1) / 2) ? n / D : (n + D - 1) / D; }

/* hierarchy: level 0 = the given sequences; tiles of D in order; repeat on the tile consensuses until one is left */
static int run_hierarchy(pseq* seqs, int ns, const pseq* backbone, const pprm* P, int D, int upper_mode, uint8_t** cons, uint32_t** cov, int want_cov) {
    pseq* cur = seqs; int ncur = ns; uint8_t** owned = NULL; int nowned = 0; int level = 0;
    *cons = NULL; if (cov) *cov = NULL;
    if (ns == 0) return 0;
    for (;;) {
        int Dl = D > 0 ? D : ncur;
        int ntiles = ntiles_of(ncur, Dl);
        pout* outs = malloc(sizeof(pout) * (size_t)(ncur + 1)); int nout = 0;
        pprm PL = *P; PL.trim_tiles = P->trim_tiles & 1; if (level > 0 && PL.trim_tiles) PL.trim_tiles |= 2;      /* upper levels: minority-insertion rule of EMIT */
        if ((P->trim_tiles & 4) && ntiles == 1) PL.trim_tiles = 0;      /* trim 3: the tile that ends a unit keeps the ends of its backbone where no member reaches them (racon's NGS windows) */
        for (int t = 0; t < ntiles; ++t) { int a = t * Dl, b = t + 1 == ntiles ? ncur : a + Dl; nout += run_tile(cur + a, b - a, backbone, &PL, outs + nout, want_cov && ntiles == 1); }
        if (getenv("ODBG_HIER") && backbone && backbone->len < atoi(getenv("ODBG_HIER"))) {      /* dev aid: tile outputs of a short (last) window, level by level */
            int nshort = 0, nlong = 0, nsemi = 0; const int wl = backbone->len;
            for (int i = 0; i < nout; ++i) { if (outs[i].a1 < wl - 1) ++nshort; if (outs[i].len > wl + 2) ++nlong; if (!(outs[i].a0 < (int)(0.01 * wl) && outs[i].a1 > wl - (int)(0.01 * wl))) ++nsemi; }
            fprintf(stderr, "[hier] level %d: %d seqs -> %d tiles -> %d outputs; span ends before the window end: %d, longer than window + 2: %d, not spanning (semi-global next): %d\n", level, ncur, ntiles, nout, nshort, nlong, nsemi);
            if (nout <= 40) for (int i = 0; i < nout; ++i) fprintf(stderr, "   out %d: len %d cw %llu span [%d, %d] tail %.*s\n", i, outs[i].len, (unsigned long long)outs[i].cw, outs[i].a0, outs[i].a1, outs[i].len < 16 ? outs[i].len : 16, (const char*)outs[i].s + (outs[i].len < 16 ? 0 : outs[i].len - 16));
        }
        if (level > 0) { free(cur); for (int i = 0; i < nowned; ++i) free(owned[i]); free(owned); owned = NULL; nowned = 0; }
        if (nout == 0) { free(outs); return 0; }
        int pick = -1;
        if (nout == 1) pick = 0;
        else if (nout >= ncur) { pick = 0; for (int i = 1; i < nout; ++i) if (outs[i].cw > outs[pick].cw) pick = i; }   /* no progress: keep the best supported */
        if (pick >= 0) {
            *cons = outs[pick].s; int len = outs[pick].len; if (cov) *cov = outs[pick].cov; else free(outs[pick].cov);
            for (int i = 0; i < nout; ++i) if (i != pick) { free(outs[i].s); free(outs[i].cov); }
            free(outs); return len;
        }
        pseq* nx = malloc(sizeof(pseq) * (size_t)nout); owned = malloc(sizeof(uint8_t*) * (size_t)nout); nowned = nout;
        for (int i = 0; i < nout; ++i) {
            uint64_t cw = outs[i].cw; int uw = cw > (1u << 20) ? (1 << 20) : (int)cw; if (uw < 1) uw = 1;
            nx[i].s = outs[i].s; nx[i].q = NULL; nx[i].len = outs[i].len; nx[i].uw = uw; nx[i].cw = (uint32_t)(cw > 0xffffffffull ? 0xffffffffull : cw); nx[i].mode = upper_mode; nx[i].a0 = 0; nx[i].a1 = -1;
            if (backbone) {   /* a tile consensus is a layer of the window like the reads it stands for: it spans [a0, a1] of the backbone and is aligned globally only if that is the whole window (racon's rule for layers, 1 % slack) */
                const int wlen = backbone->len, offset = (int)(0.01 * (double)wlen), begin = outs[i].a0, end = outs[i].a1;
                if (end >= begin) { nx[i].a0 = begin; nx[i].a1 = end; nx[i].mode = (begin < offset && end > wlen - offset) ? NGSID_POA_GLOBAL : NGSID_POA_SEMI; }
            }
            owned[i] = outs[i].s; free(outs[i].cov);
        }
        free(outs); cur = nx; ncur = nout; ++level;
    }
}

/* round 6 (ngsid_poa_params_t.single_below): a unit - a cluster in the draft, a window in the polisher - with fewer than `single_below` sequences is ONE graph in the given
   order (spoa's / racon's own order, consensus.py:257-266), with room for NGSID_POA_SINGLE_NODE_CAP / 16 times its first sequence when the longest sequence that can enter it
   (unit_maxlen: the reads of its members / behind its layers, its backbone) has at most NGSID_POA_SINGLE_MAXLEN bases; larger units are tiled at depth D */
static int run_unit(pseq* seqs, int ns, const pseq* backbone, const pprm* P, int D, int single_below, int unit_maxlen, int upper_mode, uint8_t** cons, uint32_t** cov, int want_cov) {
    if (single_below > 0 && ns < single_below) {
        pprm PS = *P; if (unit_maxlen <= NGSID_POA_SINGLE_MAXLEN) PS.node_cap = NGSID_POA_SINGLE_NODE_CAP;
        return run_hierarchy(seqs, ns, backbone, &PS, 0, upper_mode, cons, cov, want_cov);
    }
    return run_hierarchy(seqs, ns, backbone, P, D, upper_mode, cons, cov, want_cov);
}

/* library default band (band <= 0): 64 columns when every read of the call is at most 1 024 bases, else 128; the band-edge check of
   run_tile widens it per tile where a path asks for more, so the choice only decides how much work the first attempt does */
static int default_band(const ngsid_reads_t* reads) { uint64_t mx = 0; for (uint64_t i = 0; i < reads->n; ++i) { const uint64_t l = reads->off[i + 1] - reads->off[i]; if (l > mx) mx = l; } return mx <= NGSID_POA_BAND64_MAXLEN ? 64 : 128; }

static int32_t poa_consensus_impl(const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                  const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed, uint32_t* cov_out);
static const uint32_t* g_seq_weight = NULL;      /* ongsid_poa_consensus_weighted: per-read weights of the call in progress */
int32_t ongsid_poa_consensus_weighted(const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                      const ngsid_poa_params_t* prm, const uint32_t* weight, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed) {
    if (!weight) return NGSID_ERR_ARG;
    g_seq_weight = weight;
    const int32_t rc = poa_consensus_impl(reads, read_order, grp_off, n_groups, prm, cons_off, cons, cons_cap, needed, NULL);
    g_seq_weight = NULL;
    return rc;
}
int32_t ongsid_poa_consensus(const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                             const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed) {
    return poa_consensus_impl(reads, read_order, grp_off, n_groups, prm, cons_off, cons, cons_cap, needed, NULL);
}
int32_t ongsid_poa_consensus_cov(const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                 const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint32_t* cov, uint64_t cons_cap, uint64_t* needed) {
    if (!cov) return NGSID_ERR_ARG;
    return poa_consensus_impl(reads, read_order, grp_off, n_groups, prm, cons_off, cons, cons_cap, needed, cov);
}
static int32_t poa_consensus_impl(const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                  const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed, uint32_t* cov_out) {
    pprm P = { prm->match, prm->mismatch, prm->gap, prm->band > 0 ? prm->band : default_band(reads), prm->node_cap, prm->trim > 0 };
    uint64_t total = 0; int overflow = 0; cons_off[0] = 0;
    for (uint64_t g = 0; g < n_groups; ++g) {
        int ns = (int)(grp_off[g + 1] - grp_off[g]);
        pseq* seqs = malloc(sizeof(pseq) * (size_t)(ns + 1));
        for (int i = 0; i < ns; ++i) {
            uint64_t r = read_order ? read_order[grp_off[g] + (uint64_t)i] : grp_off[g] + (uint64_t)i;
            seqs[i].s = reads->seq + reads->off[r]; seqs[i].q = reads->qual ? reads->qual + reads->off[r] : NULL; seqs[i].len = (int)(reads->off[r + 1] - reads->off[r]);
            seqs[i].uw = 1; seqs[i].cw = 1; seqs[i].mode = prm->mode; seqs[i].a0 = 0; seqs[i].a1 = -1;
            if (g_seq_weight) { const uint32_t wv = g_seq_weight[r]; seqs[i].q = NULL; seqs[i].cw = wv; seqs[i].uw = (int)(wv > (1u << 20) ? (1u << 20) : (wv < 1u ? 1u : wv)); }
        }
        uint8_t* c = NULL; uint32_t* cv = NULL; int unit_maxlen = 1; for (int i = 0; i < ns; ++i) if (seqs[i].len > unit_maxlen) unit_maxlen = seqs[i].len;
        int len = run_unit(seqs, ns, NULL, &P, prm->tile_depth, prm->single_below, unit_maxlen, prm->mode, &c, cov_out ? &cv : NULL, cov_out != NULL);
        if (total + (uint64_t)len <= cons_cap) { memcpy(cons + total, c, (size_t)len); if (cov_out) for (int x = 0; x < len; ++x) cov_out[total + (uint64_t)x] = cv ? cv[x] : 0; } else overflow = 1;
        total += (uint64_t)len; cons_off[g + 1] = total;
        free(c); free(cv); free(seqs);
    }
    if (needed) *needed = total;
    if (overflow) return NGSID_ERR_CAPACITY;
    return NGSID_OK;
}

/* ---------------------------------------------------------------- racon-style polishing (consensus.py:107-126) */
int ongsid_i_hpc_minimizers(const uint8_t* s, int n, int k, int w, uint64_t* codes, uint32_t* pos);
int ongsid_i_sg_ops(const uint8_t* q, int n, const uint8_t* t, int m, int match, int mismatch, int open, int ext, uint8_t* ops);
int ongsid_i_ed_ops(const uint8_t* q, int n, const uint8_t* t, int m, uint8_t* ops);

static int cmp_u64(const void* a, const void* b) { uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b; return x < y ? -1 : (x > y); }
static int in_sorted(const uint64_t* a, int n, uint64_t key) { int lo = 0, hi = n; while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; } return lo < n && a[lo] == key; }
static uint8_t comp_base(uint8_t c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

typedef struct { pseq* v; int n, cap; } layervec;
static void lv_push(layervec* L, pseq s) { if (L->n == L->cap) { L->cap = L->cap ? L->cap * 2 : 16; L->v = realloc(L->v, sizeof(pseq) * (size_t)L->cap); } L->v[L->n++] = s; }

/* polishing windows: W bases each, a last window shorter than W/10 is merged into the one before it (see csrc/poa_host.hip polish_nwin) */
static int polish_nwin(int Bl, int W) { const int raw = Bl <= W ? 1 : (Bl + W - 1) / W; const int tail = Bl - (raw - 1) * W; return (raw >= 2 && tail < W / 10) ? raw - 1 : raw; }
static int polish_wlen(int Bl, int W, int w) { return w == polish_nwin(Bl, W) - 1 ? Bl - w * W : W; }

/* Oracle-only switches for the REFERENCE-ORDER experiments of round 5 (tools/r05_reference_order.py; the kernels implement rules = 0):
     bit 0  overlap-span clipping: minimap2 hands racon q_begin..q_end / t_begin..t_end of its chain (PAF without CIGAR: first anchor to last anchor, i.e. both ends inside an
            exact k-mer match, k = 15 for -x map-ont) and racon's edlib call aligns ONLY that span (racon src/overlap.cpp find_breaking_points).  Here: of the whole-read
            edit alignment only the columns from the first to the last run of at least 15 consecutive equal columns are kept - read ends that do not align stay out;
     bit 1  a layer that does not span its window is aligned globally to the sub-graph of its span (POA_SUBGRAPH above) instead of end-free to the whole graph;
     bit 2  PROBE, not a racon rule: unaligned head / tail bases of a window layer create no nodes (run_tile_band) - isolates the effect of source / sink nodes at the window edges. */
int32_t ongsid_debug_polish_rules(int32_t r) { const int old = g_polish_rules; if (r >= 0) g_polish_rules = r; return old; }
typedef struct { uint8_t** seq; int* len; uint64_t* used; int32_t* aln; } ptrace;      /* [it * G + g]; aln (may be NULL): include/ngsid.h ngsid_polish_trace_aln */
static int32_t polish_impl(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                           const ngsid_polish_params_t* prm, uint64_t* out_off, uint8_t* out, uint64_t out_cap, uint64_t* needed, uint64_t* n_used, ptrace* tr);
int32_t ongsid_polish(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                      const ngsid_polish_params_t* prm, uint64_t* out_off, uint8_t* out, uint64_t out_cap, uint64_t* needed, uint64_t* n_used) {
    return polish_impl(backbones, reads, read_order, grp_off, n_groups, prm, out_off, out, out_cap, needed, n_used, NULL);
}
/* the sequence after every iteration (include/ngsid.h: ngsid_polish_trace) */
static int32_t polish_trace_impl(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                            const ngsid_polish_params_t* prm, uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used, int32_t* it_aln) {
    if (!prm || !it_off || prm->iters < 1) return NGSID_ERR_ARG;
    const size_t n = (size_t)prm->iters * (size_t)n_groups;
    ptrace tr; tr.aln = it_aln; tr.seq = calloc(n + 1, sizeof(uint8_t*)); tr.len = calloc(n + 1, sizeof(int)); tr.used = calloc(n + 1, sizeof(uint64_t));
    uint64_t* ooff = calloc((size_t)n_groups + 1, sizeof(uint64_t)); uint64_t need1 = 0;
    int32_t rc = polish_impl(backbones, reads, read_order, grp_off, n_groups, prm, ooff, NULL, 0, &need1, NULL, &tr);
    free(ooff);
    if (rc == NGSID_OK || rc == NGSID_ERR_CAPACITY) {
        uint64_t total = 0; int ovf = 0; it_off[0] = 0; rc = NGSID_OK;
        for (size_t x = 0; x < n; ++x) {
            if (it_out && total + (uint64_t)tr.len[x] <= it_cap) memcpy(it_out + total, tr.seq[x], (size_t)tr.len[x]); else if (tr.len[x]) ovf = 1;
            total += (uint64_t)tr.len[x]; it_off[x + 1] = total; if (it_used) it_used[x] = tr.used[x];
        }
        if (needed) *needed = total;
        if (ovf) rc = NGSID_ERR_CAPACITY;
    }
    for (size_t x = 0; x < n; ++x) free(tr.seq[x]);
    free(tr.seq); free(tr.len); free(tr.used);
    return rc;
}
int32_t ongsid_polish_trace(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                            const ngsid_polish_params_t* prm, uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used) {
    return polish_trace_impl(backbones, reads, read_order, grp_off, n_groups, prm, it_off, it_out, it_cap, needed, it_used, NULL);
}
/* + the read -> backbone alignment of every listed read in every iteration (include/ngsid.h: ngsid_polish_trace_aln; what minimap2 writes to read_alignments_it_{i}.paf, consensus.py:112-121) */
int32_t ongsid_polish_trace_aln(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                const ngsid_polish_params_t* prm, uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used, int32_t* it_aln) {
    if (!it_aln) return NGSID_ERR_ARG;
    return polish_trace_impl(backbones, reads, read_order, grp_off, n_groups, prm, it_off, it_out, it_cap, needed, it_used, it_aln);
}
static int32_t polish_impl(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                           const ngsid_polish_params_t* prm, uint64_t* out_off, uint8_t* out, uint64_t out_cap, uint64_t* needed, uint64_t* n_used, ptrace* tr) {
    pprm P = { prm->match, prm->mismatch, prm->gap, prm->band > 0 ? prm->band : default_band(reads), prm->node_cap, (prm->trim >= 2 ? 1 : 0) | (prm->trim == 3 ? 4 : 0) };      /* trim_tiles: 1 = trim tile consensuses, 4 = but not the LAST tile of a unit (trim 3) */
    const int W = prm->window > 0 ? prm->window : 500;
    int aln_mode = prm->aln_mode;
    if (aln_mode == 2) aln_mode = 1;
    const int clip_span = (g_polish_rules & 1) || aln_mode == 3;      /* aln_mode 3 (round 5): edit distance + overlap-span clipping, see include/ngsid.h */
    if (aln_mode == 3) aln_mode = 1;
    uint64_t total = 0; int overflow = 0; out_off[0] = 0;
    for (uint64_t g = 0; g < n_groups; ++g) {
        int Blen = (int)(backbones->off[g + 1] - backbones->off[g]);
        uint8_t* B = malloc((size_t)Blen + 1); memcpy(B, backbones->seq + backbones->off[g], (size_t)Blen);
        const int ns = (int)(grp_off[g + 1] - grp_off[g]);
        /* ---- strand detection by shared (HPC) minimizers with the initial backbone: replaces minimap2's strand call */
        uint64_t* cf = malloc(sizeof(uint64_t) * (size_t)(Blen + 1)); uint64_t* cr = malloc(sizeof(uint64_t) * (size_t)(Blen + 1)); uint32_t* tp = malloc(sizeof(uint32_t) * (size_t)(Blen + 1));
        uint8_t* Brc = malloc((size_t)Blen + 1); for (int i = 0; i < Blen; ++i) Brc[i] = comp_base(B[Blen - 1 - i]);
        const int sk = prm->k < 21 ? prm->k : 21, sw = prm->w > sk ? prm->w : sk;      /* strand detection: one-word codes (comparable between sequences) */
        int nf = ongsid_i_hpc_minimizers(B, Blen, sk, sw, cf, tp); int nr = ongsid_i_hpc_minimizers(Brc, Blen, sk, sw, cr, tp);
        if (nf < 0 || nr < 0) { free(B); free(cf); free(cr); free(tp); free(Brc); return NGSID_ERR_ALPHABET; }
        qsort(cf, (size_t)nf, sizeof(uint64_t), cmp_u64); qsort(cr, (size_t)nr, sizeof(uint64_t), cmp_u64);
        int8_t* orient = malloc((size_t)ns + 1); uint8_t** rs = calloc((size_t)ns + 1, sizeof(uint8_t*)); uint8_t** rq = calloc((size_t)ns + 1, sizeof(uint8_t*)); int* rl = malloc(sizeof(int) * ((size_t)ns + 1));
        double totlen = 0.0; int rc_err = 0;
        for (int i = 0; i < ns; ++i) {
            uint64_t r = read_order ? read_order[grp_off[g] + (uint64_t)i] : grp_off[g] + (uint64_t)i; const uint8_t* s = reads->seq + reads->off[r]; const uint8_t* q = reads->qual ? reads->qual + reads->off[r] : NULL; int n = (int)(reads->off[r + 1] - reads->off[r]);
            rl[i] = n; totlen += n;
            uint64_t* c = malloc(sizeof(uint64_t) * (size_t)(n + 1)); uint32_t* p = malloc(sizeof(uint32_t) * (size_t)(n + 1));
            int cnt = ongsid_i_hpc_minimizers(s, n, sk, sw, c, p);
            if (cnt < 0) { rc_err = 1; free(c); free(p); orient[i] = -1; continue; }
            int a = 0, b = 0; for (int t = 0; t < cnt; ++t) { a += in_sorted(cf, nf, c[t]); b += in_sorted(cr, nr, c[t]); }
            orient[i] = (a == 0 && b == 0) ? -1 : (b > a ? 1 : 0);
            free(c); free(p);
            if (orient[i] < 0) continue;
            rs[i] = malloc((size_t)n + 1); rq[i] = q ? malloc((size_t)n + 1) : NULL;
            if (orient[i] == 0) { memcpy(rs[i], s, (size_t)n); if (q) memcpy(rq[i], q, (size_t)n); }
            else for (int x = 0; x < n; ++x) { rs[i][x] = comp_base(s[n - 1 - x]); if (q) rq[i][x] = q[n - 1 - x]; }
        }
        free(cf); free(cr); free(tp); free(Brc);
        if (rc_err) { for (int i = 0; i < ns; ++i) { free(rs[i]); free(rq[i]); } free(rs); free(rq); free(rl); free(orient); free(B); return NGSID_ERR_ALPHABET; }
        const int tgs = ns > 0 && (totlen / (double)ns) > 1000.0;
        uint64_t used = 0;
        for (int it = 0; it < prm->iters; ++it) {
            const int nwin = polish_nwin(Blen, W);
            layervec* LV = calloc((size_t)nwin + 1, sizeof(layervec));
            int* wmax = calloc((size_t)nwin + 1, sizeof(int));      /* single_below: the longest READ behind a window's layers */
            used = 0;
            if (tr && tr->aln) for (int i = 0; i < ns; ++i) for (int x = 0; x < 6; ++x) tr->aln[((size_t)it * (size_t)grp_off[n_groups] + (size_t)grp_off[g] + (size_t)i) * 6 + (size_t)x] = -1;
            for (int i = 0; i < ns; ++i) {
                if (orient[i] < 0) continue;
                const int n = rl[i];
                uint8_t* ops = malloc((size_t)(n + Blen + 2));
                int c = aln_mode == 1 ? ongsid_i_ed_ops(rs[i], n, B, Blen, ops)
                                           : ongsid_i_sg_ops(rs[i], n, B, Blen, prm->aln_match, prm->aln_mismatch, prm->aln_open, prm->aln_ext, ops);
                int qi = 0, ti = 0, qb = -1, tb = -1, qe = -1, te = -1, dist = 0;
                int* wf = malloc(sizeof(int) * 4 * ((size_t)nwin + 1)); for (int x = 0; x < 4 * nwin; ++x) wf[x] = -1;
                int x0 = 0, x1 = c - 1;
                if (clip_span) {        /* overlap span: first .. last run of >= 15 equal columns */
                    int run = 0, first = -1, last = -1;
                    for (int x = 0; x < c; ++x) { if (ops[x] == 0) { if (++run >= 15) { if (first < 0) first = x - 14; last = x; } } else run = 0; }
                    if (first < 0) { x0 = c; x1 = c - 1; } else { x0 = first; x1 = last; }
                }
                if (tr && tr->aln) { int a_ = 0; for (int x = 0; x < c; ++x) { if (ops[x] <= 1) { dist += ops[x]; ++a_; } else if (ops[x] == 2) { ++dist; ++a_; } else if (a_ > 0 && a_ < n) ++dist; } }      /* the distance of ongsid_ed_align_batch: the whole read, backbone ends free */
                for (int x = 0; x < c; ++x) {
                    if (x < x0 || x > x1) { if (ops[x] <= 1) { ++qi; ++ti; } else if (ops[x] == 2) ++qi; else ++ti; continue; }
                    if (ops[x] <= 1) {
                        if (qb < 0) { qb = qi; tb = ti; } qe = qi; te = ti;
                        int wdx = ti / W; if (wdx > nwin - 1) wdx = nwin - 1; if (wf[wdx * 4] < 0) { wf[wdx * 4] = qi; wf[wdx * 4 + 2] = ti; } wf[wdx * 4 + 1] = qi; wf[wdx * 4 + 3] = ti;
                        ++qi; ++ti;
                    } else if (ops[x] == 2) ++qi; else ++ti;
                }
                free(ops);
                if (tr && tr->aln) {
                    int32_t* a = tr->aln + ((size_t)it * (size_t)grp_off[n_groups] + (size_t)grp_off[g] + (size_t)i) * 6;
                    if (qb < 0) { for (int x = 0; x < 6; ++x) a[x] = -1; }
                    else { const int rcs = orient[i] == 1; a[0] = rcs; a[1] = rcs ? n - 1 - qe : qb; a[2] = rcs ? n - qb : qe + 1; a[3] = tb; a[4] = te + 1; a[5] = aln_mode == 1 ? dist : -1; }
                }
                int contributed = 0;
                if (qb >= 0) {
                    int qs = qe - qb + 1, ts = te - tb + 1; int mn = qs < ts ? qs : ts, mx = qs < ts ? ts : qs;
                    if (!(1.0 - (double)mn / (double)mx > prm->error_threshold)) {
                        for (int wdx = 0; wdx < nwin; ++wdx) {
                            if (wf[wdx * 4] < 0) continue;
                            int qf = wf[wdx * 4], ql = wf[wdx * 4 + 1], tf = wf[wdx * 4 + 2], tl = wf[wdx * 4 + 3];
                            int len = ql - qf + 1; if ((double)len < 0.02 * (double)W) continue;
                            if (rq[i]) { long sq = 0; for (int x = qf; x <= ql; ++x) sq += (long)rq[i][x] - 33; if ((double)sq / (double)len < prm->quality_threshold) continue; }
                            int ws = wdx * W, wlen = polish_wlen(Blen, W, wdx);
                            int begin = tf - ws, end = tl - ws; int offset = (int)(0.01 * (double)wlen);
                            pseq S; S.s = rs[i] + qf; S.q = rq[i] ? rq[i] + qf : NULL; S.len = len; S.uw = 1; S.cw = 1; S.a0 = begin; S.a1 = end;
                            S.mode = (begin < offset && end > wlen - offset) ? NGSID_POA_GLOBAL : ((g_polish_rules & 2) ? POA_SUBGRAPH : NGSID_POA_SEMI);
                            lv_push(&LV[wdx], S); contributed = 1; if (n > wmax[wdx]) wmax[wdx] = n;
                        }
                    }
                }
                free(wf);
                used += (uint64_t)contributed;
            }
            /* racon adds the layers of a window in the order of their first window position (src/window.cpp), ties in read order (stable) */
            for (int wdx = 0; wdx < nwin; ++wdx) {
                layervec* L = &LV[wdx];
                for (int x = 1; x < L->n; ++x) { pseq key = L->v[x]; int y = x - 1; while (y >= 0 && L->v[y].a0 > key.a0) { L->v[y + 1] = L->v[y]; --y; } L->v[y + 1] = key; }      /* insertion sort: lists are nearly sorted */
            }
            /* ---- window consensuses */
            uint8_t* NB = malloc((size_t)Blen * 3 + 1024); int nb = 0, nbcap = Blen * 3 + 1024;
            for (int wdx = 0; wdx < nwin; ++wdx) {
                int ws = wdx * W, wlen = polish_wlen(Blen, W, wdx);
                uint8_t* c = NULL; uint32_t* cov = NULL; int len = 0;
                if (LV[wdx].n >= 2) {
                    pseq bb; bb.s = B + ws; bb.q = NULL; bb.len = wlen; bb.uw = 0; bb.cw = 0; bb.mode = NGSID_POA_GLOBAL; bb.a0 = 0; bb.a1 = -1;
                    len = run_unit(LV[wdx].v, LV[wdx].n, &bb, &P, prm->tile_depth, prm->single_below, wmax[wdx] > wlen ? wmax[wdx] : (wlen > 1 ? wlen : 1), NGSID_POA_GLOBAL, &c, &cov, 1);
                    if (len > 0 && prm->trim && (tgs || prm->trim == 2) && cov) {      /* racon trims TGS windows only; trim == 2 = every window (build choice); trim == 3 = racon's window rule, tiles trimmed */
                        uint32_t avg = (uint32_t)(LV[wdx].n / 2); int b = 0, e = len - 1;
                        for (; b < len; ++b) if (cov[b] >= avg) break;
                        for (; e >= 0; --e) if (cov[e] >= avg) break;
                        if (b < e) { memmove(c, c + b, (size_t)(e - b + 1)); len = e - b + 1; }
                    }
                }
                if (len <= 0) { free(c); c = malloc((size_t)wlen + 1); memcpy(c, B + ws, (size_t)wlen); len = wlen; }
                if (nb + len > nbcap) { nbcap = (nb + len) * 2; NB = realloc(NB, (size_t)nbcap); }
                memcpy(NB + nb, c, (size_t)len); nb += len;
                free(c); free(cov); free(LV[wdx].v);
            }
            free(LV); free(wmax); free(B); B = NB; Blen = nb;
            if (tr) { const size_t x = (size_t)it * (size_t)n_groups + (size_t)g; tr->seq[x] = malloc((size_t)Blen + 1); memcpy(tr->seq[x], B, (size_t)Blen); tr->len[x] = Blen; tr->used[x] = used; }
        }
        if (n_used) n_used[g] = used;
        if (total + (uint64_t)Blen <= out_cap) memcpy(out + total, B, (size_t)Blen); else overflow = 1;
        total += (uint64_t)Blen; out_off[g + 1] = total;
        for (int i = 0; i < ns; ++i) { free(rs[i]); free(rq[i]); } free(rs); free(rq); free(rl); free(orient); free(B);
    }
    if (needed) *needed = total;
    if (overflow) return NGSID_ERR_CAPACITY;
    return NGSID_OK;
}
