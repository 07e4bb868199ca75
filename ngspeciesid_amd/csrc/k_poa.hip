// k_poa.hip - (a13,a14,a16,a17) partial-order alignment tiles: spoa-style draft consensus and racon-style window
// consensus on gfx950.  Semantics are defined (and mirrored bit for bit) by oracle/ngsid_oracle_poa.c; see its
// header for what is restated from spoa/racon and what is this build's choice.
//
// One 64-lane workgroup owns one tile (an exact-order POA of <= D sequences).  The whole graph lives in LDS
// (u16 node/edge indices: code, anchor, in/out edge lists, aligned ring, topological order+rank, edge
// tail/head/next/weight), so the inherently serial steps (traceback, heaviest bundle) run at LDS latency, and
// the data-parallel steps (banded DP row, node/edge creation, rank insertion) are lane-parallel with
// ballot/prefix-sum id allocation.  The banded DP row of a node is BW = 64*CPL columns wide: each lane owns CPL
// consecutive columns, predecessor rows come from a 16-row LDS ring (HBM copy as fallback for far predecessors),
// the in-row gap chain is a max-plus prefix scan across the wave.  Direction bytes stream to HBM row by row
// (coalesced BW-byte rows) and are pulled back 64 rows at a time into LDS for the traceback.
// Integer work throughout; the bound is LDS/VALU latency per DP row, not HBM.
#include "ngsid_internal.h"
#include "k_poa.h"
#include <algorithm>

#define PNEG (-(1 << 28))
#define SRC_SLOT 63
#define NONE16 0xFFFFu
#define HR 8

// Every graph array is an LDS (address space 3) pointer: with generic pointers the compiler emits FLAT loads, which count on
// vmcnt and therefore wait for the row's outstanding HBM stores (measured: ~2 us per DP row instead of ~0.2).
#define LDSP __attribute__((address_space(3)))
typedef LDSP uint16_t* l16; typedef LDSP uint8_t* l8; typedef LDSP int32_t* l32; typedef LDSP long long* l64;
struct G {   // LDS-resident graph of one tile + per-sequence scratch
    l16 anchor, in_first, in_last, out_first, out_last, ring, order, rank, lo, tmpv;
    l8 code;
    l16 e_tail, e_head, e_next_in, e_next_out; l32 e_w;
    l16 alnode, nodeof, ref; l8 sq;
    l32 hring; l8 dirblk; l64 sc;
    l64 rinfo; l8 rneed;                // per-rank row info for the forward pass (aliases dirblk: dead before the traceback)
};

// Single-wave workgroup: LDS instructions of one wave execute in issue order, so ordering LDS traffic between lanes only needs the
// compiler not to reorder and the LDS queue to drain - no s_barrier and, crucially, no wait on outstanding HBM stores.
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

// inclusive max-scan over the 64 lanes: 4 DPP row shifts inside the 16-lane rows, row totals through readlane (SGPRs)
__device__ __forceinline__ int wave_incl_max_scan(int v, int lane, int ident)
{
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int add = lane >= 48 ? max(r0, max(r1, r2)) : (lane >= 32 ? max(r0, r1) : (lane >= 16 ? r0 : ident));
    return max(v, add);
}

struct TS { int V, E, L0, members, nout, capV, capE; unsigned long long cw_sum; };

__device__ __forceinline__ int wtof(const PSeq& S, int i) { return S.q ? (int)S.q[i] - 33 : S.uw; }

__device__ __forceinline__ int band_lo(int anchor, const PSeq& S, int L0, int BW) {
    int a0 = S.a0, a1 = S.a1; if (a1 < a0) { a0 = 0; a1 = L0 - 1; }
    long long span = (long long)a1 - a0 + 1; if (span < 1) span = 1;
    long long c = ((long long)(anchor - a0) * (long long)S.len) / span;
    long long lo = c - BW / 2; long long mx = (long long)S.len + 1 - BW; if (mx < 0) mx = 0;
    if (lo < 0) lo = 0; if (lo > mx) lo = mx;
    return (int)lo;
}

__device__ void tile_add_first(const G& g, uint32_t* cov, const PSeq& S, TS& st, int lane)
{
    for (int i = lane; i < S.len; i += 64) {
        g.code[i] = S.s[i]; g.anchor[i] = (uint16_t)i; g.ring[i] = (uint16_t)i; g.order[i] = (uint16_t)i; g.rank[i] = (uint16_t)i; cov[i] = S.cw;
        g.in_first[i] = g.in_last[i] = (i > 0) ? (uint16_t)(i - 1) : (uint16_t)NONE16;
        g.out_first[i] = g.out_last[i] = (i + 1 < S.len) ? (uint16_t)i : (uint16_t)NONE16;
        if (i > 0) { const int e = i - 1; g.e_tail[e] = (uint16_t)(i - 1); g.e_head[e] = (uint16_t)i; g.e_next_in[e] = NONE16; g.e_next_out[e] = NONE16; g.e_w[e] = wtof(S, i - 1) + wtof(S, i); }
    }
    st.V = S.len; st.E = S.len > 0 ? S.len - 1 : 0; st.L0 = S.len; st.cw_sum += S.cw;
    __threadfence_block();
    __syncthreads();
}

// heaviest bundle + branch completion (oracle g_consensus); lane 0, everything in LDS
__device__ void tile_emit(const G& g, const uint32_t* cov, const PoaJobSet& J, uint32_t job, TS& st, int lane)
{
    if (st.V == 0 || st.members == 0) return;
    if (st.nout >= J.D) { if (lane == 0 && J.slot_overflow) atomicExch(J.slot_overflow, 1u); return; }   // host retries with more output slots
    const size_t slot = (size_t)job * J.D + st.nout;
    uint8_t* dst = J.out + slot * (size_t)J.Vcap;
    uint32_t* dcov = J.out_cov ? J.out_cov + slot * (size_t)J.Vcap : nullptr;
    const int V = st.V;
    __threadfence_block();
    __syncthreads();
    if (lane == 0) {
        l16 pred = g.lo;
        int mx = -1;
        for (int r = 0; r < V; ++r) {
            const int v = g.order[r]; long long sv = -1; int pv = NONE16;
            for (int e = g.in_first[v]; e != NONE16; e = g.e_next_in[e]) {
                const int t = g.e_tail[e]; const long long w = g.e_w[e];
                if (sv < w || (sv == w && g.sc[pv] <= g.sc[t])) { sv = w; pv = t; }
            }
            if (pv != NONE16) sv += g.sc[pv];
            g.sc[v] = sv; pred[v] = (uint16_t)pv;
            if (mx < 0 || g.sc[mx] < sv) mx = v;
        }
        while (g.out_first[mx] != NONE16) {
            const int start = mx;
            for (int e = g.out_first[start]; e != NONE16; e = g.e_next_out[e])
                for (int f = g.in_first[g.e_head[e]]; f != NONE16; f = g.e_next_in[f]) if (g.e_tail[f] != start) g.sc[g.e_tail[f]] = -1;
            int m2 = -1;
            for (int r = g.rank[start] + 1; r < V; ++r) {
                const int v = g.order[r]; long long sv = -1; int pv = NONE16;
                for (int e = g.in_first[v]; e != NONE16; e = g.e_next_in[e]) {
                    const int t = g.e_tail[e]; if (g.sc[t] == -1) continue; const long long w = g.e_w[e];
                    if (sv < w || (sv == w && g.sc[pv] <= g.sc[t])) { sv = w; pv = t; }
                }
                if (pv != NONE16) sv += g.sc[pv];
                g.sc[v] = sv; pred[v] = (uint16_t)pv;
                if (m2 < 0 || g.sc[m2] < sv) m2 = v;
            }
            if (m2 < 0) break;
            mx = m2;
        }
        int n = 0; for (int v = mx; v != NONE16; v = pred[v]) ++n;
        int i = n; for (int v = mx; v != NONE16; v = pred[v]) { --i; dst[i] = g.code[v]; if (dcov) { uint32_t c = cov[v]; for (int u = g.ring[v]; u != v; u = g.ring[u]) c += cov[u]; dcov[i] = c; } }
        if (J.trim_tiles && dcov && n > 0) {   // oracle EMIT: coverage-trim the tile consensus ends
            const uint32_t thr = (uint32_t)(st.cw_sum / 2); int b = 0, e = n - 1;
            for (; b < n; ++b) if (dcov[b] >= thr) break;
            for (; e >= 0; --e) if (dcov[e] >= thr) break;
            if (b < e) { const int m2 = e - b + 1; if (b > 0) for (int x = 0; x < m2; ++x) { dst[x] = dst[b + x]; dcov[x] = dcov[b + x]; } n = m2; }
        }
        J.out_len[slot] = n; J.out_cw[slot] = st.cw_sum;
    }
    __syncthreads();
    st.nout += 1;
}

// align S to the graph and merge it.  returns 0 = dropped (no valid end cell), 1 = added, 2 = does not fit
template <int CPL>
__device__ int tile_align_add(const G& g, uint32_t* cov, int32_t* Hg, uint8_t* Dg, const PoaJobSet& J, const PSeq& S, TS& st, int lane)
{
    constexpr int BW = 64 * CPL;
    const int L = S.len, mode = S.mode, gp = J.g, V = st.V;
    // ---------- per-rank row info, built lane-parallel so that the serial row loop reads ONE 8-byte LDS word per row:
    //   lo:16 | first pred rank:16 | second pred rank:16 | letter:8 | flags:8
    //   flags: 1 no predecessor, 2 more than two, 4 sink, 8 keep an HBM copy (a successor is > HR rows away),
    //          16 chain row (single predecessor = previous row, band shift 0/1), 32 = that band shift
    for (int r = lane; r < V; r += 64) g.rneed[r] = 0;
    lds_sync();
    for (int r = lane; r < V; r += 64) {
        const int v = g.order[r];
        for (int e = g.in_first[v]; e != NONE16; e = g.e_next_in[e]) { const int pr = g.rank[g.e_tail[e]]; if (r - pr > HR) g.rneed[pr] = 1; }
        const int l0 = band_lo(g.anchor[v], S, st.L0, BW);
        g.lo[r] = (uint16_t)l0;
        const int e0 = g.in_first[v]; int p0 = NONE16, p1 = NONE16, fl = 0;
        if (e0 == NONE16) fl |= 1;
        else { p0 = g.rank[g.e_tail[e0]]; const int e1 = g.e_next_in[e0]; if (e1 != NONE16) { p1 = g.rank[g.e_tail[e1]]; if (g.e_next_in[e1] != NONE16) fl |= 2; } }
        if (g.out_first[v] == NONE16) fl |= 4;
        if (r > 0 && !(fl & 3) && p0 == r - 1 && p1 == NONE16) {
            const int d = l0 - band_lo(g.anchor[g.order[r - 1]], S, st.L0, BW);
            if (d == 0 || d == 1) fl |= 16 | (d << 5);
        }
        g.rinfo[r] = (unsigned long long)(unsigned)l0 | ((unsigned long long)(unsigned)p0 << 16) | ((unsigned long long)(unsigned)p1 << 32)
                   | ((unsigned long long)g.code[v] << 48) | ((unsigned long long)(unsigned)fl << 56);
    }
    for (int i = lane; i < L; i += 64) { g.alnode[i] = NONE16; g.sq[i] = S.s[i]; }
    lds_sync();
    for (int r = lane; r < V; r += 64) if (g.rneed[r]) g.rinfo[r] |= 8ull << 56;
    __syncthreads();
    // ---------- forward DP, one row per graph node in topological order
    const bool has_invalid = (L + 1 < BW);              // otherwise every band column is <= L (band_lo clamps)
    const bool local = mode == NGSID_POA_LOCAL, semi = mode == NGSID_POA_SEMI;
    const int sm = J.m, sn = J.n;
    int bestv = PNEG, bestpk = 0x7fffffff;               // packed (rank << 8 | band column): ties -> lowest rank, then lowest column
    int hprev[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) hprev[c] = PNEG;
    unsigned long long ri_next = V > 0 ? g.rinfo[0] : 0ull;
    uint8_t* dgp = Dg + lane * CPL;
    for (int r = 0; r < V; ++r, dgp += BW) {
        const unsigned rlo = __builtin_amdgcn_readfirstlane((unsigned)ri_next), rhi = __builtin_amdgcn_readfirstlane((unsigned)(ri_next >> 32));
        if (r + 1 < V) ri_next = g.rinfo[r + 1];          // prefetch: consumed one iteration later
        const int l0 = rlo & 0xffff, p0r = rlo >> 16, p1r = rhi & 0xffff; const int cv = (rhi >> 16) & 0xff; const int rfl = rhi >> 24;
        const int jb = l0 + lane * CPL;
        int scj[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) { const int j = jb + c; const int ch = (j >= 1 && j <= L) ? (int)g.sq[j - 1] : 256; scj[c] = (ch == cv) ? sm : sn; }
        int X[CPL], Dd[CPL];
        if ((rfl & 16) && !semi) {
            // chain row: the only predecessor is the previous row, still in registers; neighbours through one DPP move
            const int lf = __builtin_amdgcn_update_dpp(PNEG, hprev[CPL - 1], 0x138, 0xf, 0xf, false);   // lane-1's last column
            const int rt = __builtin_amdgcn_update_dpp(PNEG, hprev[0], 0x130, 0xf, 0xf, false);         // lane+1's first column
            int ext[CPL + 2]; ext[0] = lf; ext[CPL + 1] = rt;
#pragma unroll
            for (int c = 0; c < CPL; ++c) ext[c + 1] = hprev[c];
            const bool sh = (rfl & 32) != 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int up = sh ? ext[c + 2] : ext[c + 1]; int dg = sh ? ext[c + 1] : ext[c];
                if (jb + c < 1) dg = PNEG;
                const int xu = up + gp, xd = dg + scj[c];
                X[c] = xd >= xu ? xd : xu; Dd[c] = xd >= xu ? 0 : 1;
            }
        } else {
            const bool nopred = (rfl & 1) != 0;
            const bool use_src = nopred || semi;
            int Xd[CPL], Dslot[CPL], Xu[CPL], Uslot[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) { Xd[c] = PNEG; Xu[c] = PNEG; Dslot[c] = 0; Uslot[c] = 0; }
            int slot = 0, eit = NONE16;
            if (rfl & 2) { eit = g.in_first[g.order[r]]; }
            for (;; ++slot) {
                int pr;
                if (!(rfl & 2)) { pr = slot == 0 ? p0r : (slot == 1 ? p1r : NONE16); if (pr == NONE16) break; }
                else { if (eit == NONE16) break; pr = g.rank[g.e_tail[eit]]; eit = g.e_next_in[eit]; }
                const int plo = g.lo[pr];
                const int pc0 = jb - plo;
                int hp[CPL + 1];
                if ((r - pr) <= HR) {                       // LDS ring
                    const l32 Hp = g.hring + (size_t)(pr & (HR - 1)) * BW;
#pragma unroll
                    for (int c = 0; c <= CPL; ++c) { const int pc = pc0 - 1 + c; hp[c] = (pc >= 0 && pc < BW) ? Hp[pc] : PNEG; }
                } else {                                    // far predecessor: HBM copy of the row (flag 8 made the producer store it)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int32_t* Hq = Hg + (size_t)pr * BW;
#pragma unroll
                    for (int c = 0; c <= CPL; ++c) { const int pc = pc0 - 1 + c; hp[c] = (pc >= 0 && pc < BW) ? __builtin_nontemporal_load(Hq + pc) : PNEG; }
                }
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int j = jb + c;
                    { const int hv = hp[c + 1]; if (hv > PNEG && hv + gp > Xu[c]) { Xu[c] = hv + gp; Uslot[c] = slot; } }
                    if (j >= 1) { const int hv = hp[c]; if (hv > PNEG && hv + scj[c] > Xd[c]) { Xd[c] = hv + scj[c]; Dslot[c] = slot; } }
                }
            }
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = jb + c;
                if (use_src && j >= 1) { const int sv = local ? 0 : (j - 1) * gp; if (sv + scj[c] > Xd[c]) { Xd[c] = sv + scj[c]; Dslot[c] = SRC_SLOT; } }
                if (nopred && !semi) { const int sv = local ? 0 : j * gp; if (sv + gp > Xu[c]) { Xu[c] = sv + gp; Uslot[c] = SRC_SLOT; } }
                if (Xd[c] >= Xu[c]) { X[c] = Xd[c]; Dd[c] = 0 | (Dslot[c] << 2); } else { X[c] = Xu[c]; Dd[c] = 1 | (Uslot[c] << 2); }
            }
        }
        // in-row gap chain H[j] = max(Xf[j], H[j-1]+g) as a max-plus prefix scan of y[j] = Xf[j] - j*g
        int exl[CPL]; int run = PNEG * 2;
        const int jg0 = jb * gp;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            if (X[c] < PNEG / 2) { X[c] = PNEG; Dd[c] = 3; }                   // unreachable (sentinel arithmetic may have drifted)
            if (has_invalid && jb + c > L) { X[c] = PNEG; Dd[c] = 3; }
            const int xf = (local && X[c] < 0) ? 0 : X[c];
            int y = xf - (jg0 + c * gp);
            if (has_invalid && jb + c > L) y = PNEG * 2;
            exl[c] = run; run = max(run, y);
        }
        const int incl = wave_incl_max_scan(run, lane, PNEG * 2);
        const int excl_lane = __builtin_amdgcn_update_dpp(PNEG * 2, incl, 0x138, 0xf, 0xf, false);     // wave_shr:1, lane 0 keeps the identity
        int hrow[CPL]; unsigned dpack = 0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int ex = max(excl_lane, exl[c]);
            int val = X[c], dd = Dd[c];
            const int lfv = ex + jg0 + c * gp;                  // best value reachable through the in-row gap chain
            if (lfv > val && lfv > PNEG / 2) { val = lfv; dd = 2; }
            if (local && val <= 0) { val = 0; dd = 3; }
            if (val <= PNEG / 2) { val = PNEG; dd = 3; }
            if (has_invalid && jb + c > L) { val = PNEG; dd = 3; }
            hrow[c] = val; hprev[c] = val; dpack |= (unsigned)(dd & 0xff) << (8 * c);
        }
        if (local) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) if (hrow[c] > bestv) { bestv = hrow[c]; bestpk = (r << 8) | (lane * CPL + c); }
        } else if (semi || (rfl & 4)) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) if (jb + c == L && hrow[c] > PNEG && hrow[c] > bestv) { bestv = hrow[c]; bestpk = (r << 8) | (lane * CPL + c); }
        }
        // publish the row: LDS ring for the next rows, HBM copy only where a far successor will ask for it, packed direction bytes
        l32 ring = g.hring + (size_t)(r & (HR - 1)) * BW + lane * CPL;
#pragma unroll
        for (int c = 0; c < CPL; ++c) ring[c] = hrow[c];
        if (rfl & 8) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) Hg[(size_t)r * BW + lane * CPL + c] = hrow[c];
        }
        if (CPL == 1) *dgp = (uint8_t)dpack; else if (CPL == 2) *(uint16_t*)dgp = (uint16_t)dpack; else *(unsigned int*)dgp = dpack;
        lds_sync();                                   // next row may read this ring slot; HBM stores stay in flight
    }
    __threadfence_block();                            // direction rows must have landed before the traceback pulls them back
    __syncthreads();
    // ---------- best end cell: max value, ties -> lowest rank, then lowest column
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int ov = __shfl_xor(bestv, d), opk = __shfl_xor(bestpk, d);
        if (ov > bestv || (ov == bestv && opk < bestpk)) { bestv = ov; bestpk = opk; }
    }
    const int bestr = bestpk == 0x7fffffff ? -1 : (bestpk >> 8), bestc = bestpk & 0xff;
    bool aligned_any = true;
    if (bestr < 0 || (mode == NGSID_POA_LOCAL && bestv <= 0)) {
        if (mode != NGSID_POA_LOCAL) return 0;
        aligned_any = false;                                   // nothing aligned: the whole read becomes a new branch
    }
    // ---------- traceback (uniform across lanes; direction rows pulled 64 at a time into LDS)
    if (aligned_any) {
        int r = bestr, c = bestc, j = g.lo[bestr] + bestc;
        int blk_hi = -1, blk_lo = 0;
        for (;;) {
            if (r > blk_hi || r < blk_lo) {
                __syncthreads();
                blk_hi = r; blk_lo = r - 63 < 0 ? 0 : r - 63;
                const int rr = blk_hi - lane;
                if (rr >= blk_lo) { const uint8_t* src = Dg + (size_t)rr * BW; l8 dstp = g.dirblk + (size_t)lane * BW; for (int x = 0; x < BW; x += 16) *(LDSP ngsid_v4u*)(dstp + x) = ngsid_load16_l2(src + x); }   // L2-served: scratch rows are rewritten per sequence
                __threadfence_block();
                __syncthreads();
            }
            const int v = g.order[r]; const int d = g.dirblk[(size_t)(blk_hi - r) * BW + c]; const int type = d & 3, slot = d >> 2;
            if (type == 3) break;
            if (type == 2) { --j; --c; continue; }
            if (type == 0) { if (lane == 0) g.alnode[j - 1] = (uint16_t)v; --j; }
            if (slot == SRC_SLOT) break;
            int e = g.in_first[v]; for (int t = 0; t < slot; ++t) e = g.e_next_in[e];
            r = g.rank[g.e_tail[e]]; c = j - g.lo[r];
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---------- A: existing node per position (same letter on the aligned node or one of its siblings)
    int nnew = 0;
    for (int i0 = 0; i0 < L; i0 += 64) {
        const int i = i0 + lane; bool isnew = false;
        if (i < L) {
            const int v = g.alnode[i]; const uint8_t ch = g.sq[i]; int found = NONE16;
            if (v != NONE16) { if (g.code[v] == ch) found = v; else for (int u = g.ring[v]; u != v; u = g.ring[u]) if (g.code[u] == ch) { found = u; break; } }
            g.nodeof[i] = (uint16_t)found; isnew = found == NONE16;
        }
        nnew += __popcll(__ballot(isnew));
    }
    if (V + nnew > st.capV || st.E + L > st.capE) { __syncthreads(); return 2; }      // oracle g_add_alignment capacity rule
    // ---------- B: ref(i) = aligned node of the first aligned position >= i (reverse carry scan)
    {
        int carry = NONE16;
        for (int i0 = ((L - 1) / 64) * 64; i0 >= 0; i0 -= 64) {
            const int i = i0 + lane; const int a = (i < L) ? g.alnode[i] : NONE16;
            const unsigned long long m = __ballot(a != NONE16);
            const unsigned long long ge = m & (~0ull << lane);
            const int src = ge ? __ffsll((long long)ge) - 1 : 0;
            const int val = __shfl(a, src);
            if (i < L) g.ref[i] = (uint16_t)(ge ? val : carry);
            if (m) { const int first = __ffsll((long long)m) - 1; carry = __shfl(a, first); }
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---------- C: create nodes (ids in sequence order).  anchor: nearest aligned position at or before i, else after, else a0.
    //             tmpv[k] = old rank the k-th new node is inserted before (V = end); non-decreasing in k.
    {
        int base = V, lastal = NONE16;
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane; const int a = (i < L) ? g.alnode[i] : NONE16; const bool isnew = (i < L) && g.nodeof[i] == NONE16;
            const unsigned long long ma = __ballot(a != NONE16);
            const unsigned long long le = ma & (~0ull >> (63 - lane));
            const int src = le ? 63 - __clzll(le) : 0;
            const int lv = __shfl(a, src);
            const int la = le ? lv : lastal;
            const unsigned long long mn = __ballot(isnew);
            const int before = __popcll(mn & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
            if (isnew) {
                const int y = base + before; const int rf = g.ref[i];
                const int anc = la != NONE16 ? g.anchor[la] : (rf != NONE16 ? g.anchor[rf] : (S.a1 < S.a0 ? 0 : S.a0));
                g.code[y] = g.sq[i]; g.anchor[y] = (uint16_t)anc; g.in_first[y] = g.in_last[y] = g.out_first[y] = g.out_last[y] = NONE16; cov[y] = 0;
                if (a != NONE16) { g.ring[y] = g.ring[a]; g.ring[a] = (uint16_t)y; } else g.ring[y] = (uint16_t)y;
                g.nodeof[i] = (uint16_t)y;
                g.tmpv[y - V] = (uint16_t)(rf != NONE16 ? g.rank[rf] : V);
            }
            base += __popcll(mn);
            if (ma) { const int hl = 63 - __clzll(ma); lastal = __shfl(a, hl); }
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---------- D: ranks.  k-th new node -> tmpv[k] + k ; old node at rank p -> p + #{k : tmpv[k] <= p}
    {
        l16 neworder = g.lo;
        for (int p = lane; p < V; p += 64) {
            int lo = 0, hi = nnew; while (lo < hi) { const int mid = (lo + hi) >> 1; if (g.tmpv[mid] <= p) lo = mid + 1; else hi = mid; }
            neworder[p + lo] = g.order[p];
        }
        for (int k = lane; k < nnew; k += 64) neworder[g.tmpv[k] + k] = (uint16_t)(V + k);
        __threadfence_block();
        __syncthreads();
        for (int r = lane; r < V + nnew; r += 64) { const int v = neworder[r]; g.order[r] = (uint16_t)v; g.rank[v] = (uint16_t)r; }
    }
    __threadfence_block();
    __syncthreads();
    // ---------- E: coverage and edges (edge ids in sequence order)
    {
        int ebase = st.E;
        for (int i0 = 0; i0 < L; i0 += 64) {
            const int i = i0 + lane; bool newedge = false; int a = 0, b = 0, w = 0;
            if (i < L) {
                b = g.nodeof[i]; cov[b] += S.cw;
                if (i > 0) {
                    a = g.nodeof[i - 1]; w = wtof(S, i - 1) + wtof(S, i);
                    int e = g.out_first[a];
                    for (; e != NONE16; e = g.e_next_out[e]) if (g.e_head[e] == b) break;
                    if (e != NONE16) g.e_w[e] += w; else newedge = true;
                }
            }
            const unsigned long long mn = __ballot(newedge);
            if (newedge) {
                const int e = ebase + __popcll(mn & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                g.e_tail[e] = (uint16_t)a; g.e_head[e] = (uint16_t)b; g.e_w[e] = w; g.e_next_in[e] = NONE16; g.e_next_out[e] = NONE16;
                if (g.out_last[a] == NONE16) g.out_first[a] = (uint16_t)e; else g.e_next_out[g.out_last[a]] = (uint16_t)e; g.out_last[a] = (uint16_t)e;
                if (g.in_last[b] == NONE16) g.in_first[b] = (uint16_t)e; else g.e_next_in[g.in_last[b]] = (uint16_t)e; g.in_last[b] = (uint16_t)e;
            }
            ebase += __popcll(mn);
            lds_sync();
        }
        st.E = ebase;
    }
    st.V = V + nnew; st.cw_sum += S.cw;
    __threadfence_block();
    __syncthreads();
    return 1;
}

template <int CPL>
__global__ __launch_bounds__(64)
void k_poa_tile(PoaJobSet J)
{
    constexpr int BW = 64 * CPL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int Vc = J.Vcap, Ec = J.Ecap, Lm = J.Lmax;
    G g;
    {
        LDSP unsigned char* base = (LDSP unsigned char*)smem;
        size_t o = 0;
        auto take = [&](size_t bytes) { LDSP unsigned char* q = base + o; o += (bytes + 15) & ~(size_t)15; return q; };
        g.anchor = (l16)take(2 * Vc); g.in_first = (l16)take(2 * Vc); g.in_last = (l16)take(2 * Vc); g.out_first = (l16)take(2 * Vc);
        g.out_last = (l16)take(2 * Vc); g.ring = (l16)take(2 * Vc); g.order = (l16)take(2 * Vc); g.rank = (l16)take(2 * Vc);
        g.lo = (l16)take(2 * (Vc + 1)); g.tmpv = (l16)take(2 * (Vc + 1)); g.code = (l8)take(Vc);
        g.e_tail = (l16)take(2 * Ec); g.e_head = (l16)take(2 * Ec); g.e_next_in = (l16)take(2 * Ec); g.e_next_out = (l16)take(2 * Ec); g.e_w = (l32)take(4 * Ec);
        g.alnode = (l16)take(2 * Lm); g.nodeof = (l16)take(2 * Lm); g.ref = (l16)take(2 * Lm); g.sq = (l8)take(Lm);
        size_t blk = (size_t)64 * BW; if (blk < (size_t)9 * Vc + 32) blk = ((size_t)9 * Vc + 32 + 15) & ~(size_t)15;
        size_t dp = (size_t)HR * BW * 4 + blk; if (dp < (size_t)8 * Vc) dp = (size_t)8 * Vc;
        LDSP unsigned char* dpr = take(dp);
        g.hring = (l32)dpr; g.dirblk = (l8)(dpr + (size_t)HR * BW * 4); g.sc = (l64)dpr;
        g.rinfo = (l64)g.dirblk; g.rneed = (l8)(g.rinfo + Vc);
    }
    int32_t* Hg = J.Hglob + (size_t)blockIdx.x * Vc * BW;
    uint8_t* Dg = J.dirglob + (size_t)blockIdx.x * Vc * BW;
    uint32_t* cov = J.covglob + (size_t)blockIdx.x * Vc;

    for (uint32_t job = blockIdx.x; job < J.njobs; job += gridDim.x) {
        const uint32_t s0 = J.job_off[job], s1 = J.job_off[job + 1];
        const int bbi = J.job_bb ? J.job_bb[job] : -1;
        TS st; st.V = 0; st.E = 0; st.L0 = 0; st.members = 0; st.nout = 0; st.cw_sum = 0;
        uint32_t ndrop = 0;
        {   // per-job capacity = oracle run_tile: cap_for(L0) but at least the longest member + 1; edges 2x
            int maxlen = bbi >= 0 ? J.bbs[bbi].len : 0, first = bbi >= 0 ? J.bbs[bbi].len : 0;
            for (uint32_t si = s0; si < s1; ++si) { const int l = J.seqs[J.seq_idx ? J.seq_idx[si] : si].len; if (l > maxlen) maxlen = l; if (first == 0 && bbi < 0 && si == s0) first = l; }
            long long c = (long long)(first > 0 ? first : 1) * (J.node_cap > 0 ? J.node_cap : 28) / 16; if (c < (first > 0 ? first : 1) + 64) c = (first > 0 ? first : 1) + 64;
            if (c < maxlen + 1) c = maxlen + 1;
            st.capV = (int)(c < Vc ? c : Vc); st.capE = 3 * st.capV / 2 < Ec ? 3 * st.capV / 2 : Ec;
        }
        for (uint32_t si = s0; si < s1; ++si) {
            const PSeq S = J.seqs[J.seq_idx ? J.seq_idx[si] : si];
            if (S.len <= 0) continue;
            if (S.len > Lm) { ++ndrop; continue; }
            if (st.V == 0) {
                if (bbi >= 0) { const PSeq B = J.bbs[bbi]; tile_add_first(g, cov, B, st, lane); }
                else { if (S.len > st.capV) { ++ndrop; continue; } tile_add_first(g, cov, S, st, lane); st.members = 1; continue; }
            }
            int rcode = tile_align_add<CPL>(g, cov, Hg, Dg, J, S, st, lane);
            if (rcode == 0) { ++ndrop; continue; }
            if (rcode == 2) {
                tile_emit(g, cov, J, job, st, lane);
                st.V = 0; st.E = 0; st.L0 = 0; st.members = 0; st.cw_sum = 0;
                if (bbi >= 0) { const PSeq B = J.bbs[bbi]; tile_add_first(g, cov, B, st, lane); rcode = tile_align_add<CPL>(g, cov, Hg, Dg, J, S, st, lane); if (rcode == 1) st.members = 1; else ++ndrop; }
                else if (S.len <= st.capV) { tile_add_first(g, cov, S, st, lane); st.members = 1; } else ++ndrop;
                continue;
            }
            st.members += 1;
        }
        tile_emit(g, cov, J, job, st, lane);
        if (lane == 0) { J.out_n[job] = (uint32_t)st.nout; if (ndrop && J.dropped) atomicAdd(J.dropped, ndrop); }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ host side
size_t poa_lds_bytes(int Vc, int Ec, int Lm, int BW)
{
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    size_t t = 8 * al(2 * (size_t)Vc) + 2 * al(2 * ((size_t)Vc + 1)) + al(Vc) + 4 * al(2 * (size_t)Ec) + al(4 * (size_t)Ec) + 3 * al(2 * (size_t)Lm) + al(Lm);
    size_t blk = (size_t)64 * BW; if (blk < (size_t)9 * Vc + 32) blk = ((size_t)9 * Vc + 32 + 15) & ~(size_t)15;
    size_t dp = (size_t)HR * BW * 4 + blk; if (dp < (size_t)8 * Vc) dp = (size_t)8 * Vc;
    return t + al(dp);
}

int32_t poa_run_jobs(ngsid_ctx* ctx, PoaJobSet J, int band)
{
    if (J.njobs == 0) return NGSID_OK;
    const int BW = band <= 64 ? 64 : (band <= 128 ? 128 : 256);
    if (J.g >= 0) NGSID_FAIL(ctx, NGSID_ERR_ARG, "POA gap score must be negative");
    if (J.Vcap > 0xFFF0 || J.Ecap > 0xFFF0) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA graph capacity exceeds 16-bit indices (sequence too long for the LDS-resident tile)");
    const size_t lds = poa_lds_bytes(J.Vcap, J.Ecap, J.Lmax, BW);
    if (lds > 160 * 1024) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA tile needs %zu bytes of LDS (> 160 KiB): sequences too long", lds);
    const int per_cu = std::max<int>(1, (int)((160 * 1024) / lds));
    uint32_t nwg = (uint32_t)std::min<uint64_t>(J.njobs, (uint64_t)ctx->n_cu * std::min(per_cu, 8));
    const size_t cells = (size_t)J.Vcap * BW;
    if (ctx->poa_h.n < nwg * cells) HIPCHK(ctx, ctx->poa_h.alloc(nwg * cells));
    if (ctx->poa_d.n < nwg * cells) HIPCHK(ctx, ctx->poa_d.alloc(nwg * cells));
    if (ctx->poa_cov.n < (size_t)nwg * J.Vcap) HIPCHK(ctx, ctx->poa_cov.alloc((size_t)nwg * J.Vcap));
    J.Hglob = ctx->poa_h.p; J.dirglob = ctx->poa_d.p; J.covglob = ctx->poa_cov.p;
    if (BW == 64) { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ProfScope ps_(ctx, "k_poa_tile"); hipLaunchKernelGGL(k_poa_tile<1>, dim3(nwg), dim3(64), lds, ctx->stream, J); }
    else if (BW == 128) { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ProfScope ps_(ctx, "k_poa_tile"); hipLaunchKernelGGL(k_poa_tile<2>, dim3(nwg), dim3(64), lds, ctx->stream, J); }
    else { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); ProfScope ps_(ctx, "k_poa_tile"); hipLaunchKernelGGL(k_poa_tile<4>, dim3(nwg), dim3(64), lds, ctx->stream, J); }
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}
