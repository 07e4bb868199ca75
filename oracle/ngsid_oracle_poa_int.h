/* ngsid_oracle_poa_int.h - types shared by the two tile engines of the CPU oracle (TEST INFRASTRUCTURE ONLY, see ngsid_oracle.h) */
#ifndef NGSID_ORACLE_POA_INT_H
#define NGSID_ORACLE_POA_INT_H
#include <stdint.h>
#include "ngsid_oracle.h"

#define PNEG (-(1 << 28))
#define SRC_SLOT 63

/* one sequence handed to the tile engine */
typedef struct {
    const uint8_t* s; const uint8_t* q; int len;   /* q == NULL: uniform weight uw */
    int uw; uint32_t cw; int mode; int a0, a1;     /* a1 < a0: span = whole first sequence */
} pseq;
static inline int wt(const pseq* S, int i) { return S->q ? (int)S->q[i] - 33 : S->uw; }
typedef struct { int node, pos; } ppair;
typedef struct { uint8_t* s; uint32_t* cov; int len; uint64_t cw; int a0, a1; /* anchors (coordinates in the first sequence of the graph) of the first / last consensus node */ } pout;
typedef struct { int m, n, g, band, node_cap, trim_tiles; } pprm;
static inline int cap_for(int L0, int node_cap) { long c = (long)L0 * (node_cap > 0 ? node_cap : 28) / 16; if (c < L0 + 64) c = L0 + 64; return (int)c; }

/* ngsid_oracle_poa_rank.c: the same tile engine on a rank-ordered graph (the representation of csrc/k_poa.hip since round 4) */
int run_tile_band_rank(const pseq* seqs, int ns, const pseq* backbone, const pprm* P, int band, int* edge, pout* outs, int want_cov);
#endif
