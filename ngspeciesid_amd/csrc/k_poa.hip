// k_poa.hip - (a13,a14,a16,a17) partial-order alignment tiles: spoa-style draft consensus and racon-style window
// consensus on gfx950.  Semantics are defined (and mirrored bit for bit) by oracle/ngsid_oracle_poa.c; see its
// header for what is restated from spoa/racon and what is this build's choice.
//
// One 64-lane workgroup owns one tile (an exact-order POA of <= D sequences).
//   * The GRAPH (node letters, anchors, in/out edge lists, aligned rings, topological order+rank, edge weights, coverage;
//     u16 indices) lives in a per-workgroup HBM scratch that stays L2 resident; it is only touched by lane-parallel
//     phases (row-info build, node matching, node/edge creation, rank insertion) where 64 independent requests hide
//     the L2 latency.
//   * The per-ALIGNMENT working set lives in LDS (~20-26 KB, so 6-8 tiles are resident per CU): one packed 8-byte
//     "row info" word per topological rank (band start, ranks of the first two predecessors, letter, flags), an
//     8-row ring of DP rows, the sequence, a 32-row block of direction bytes for the traceback, and the per-position
//     alignment result.  The serial loops (DP rows, traceback, heaviest bundle) run entirely out of LDS/registers.
//   * DP row of a node: BW = 64*CPL band columns, CPL per lane.  Chain rows (single predecessor = previous row, ~85 %)
//     take their inputs from registers + one DPP lane shift; other rows read predecessor rows from the LDS ring
//     (HBM copy for predecessors more than 8 rows back).  The in-row gap chain is a DPP max-plus prefix scan.
//     Direction bytes stream to HBM (coalesced) and come back 32 rows at a time for the traceback.
// Integer work throughout; the bound is instruction issue / LDS latency per DP row at one wave per tile, not HBM.
#include "ngsid_internal.h"
#include "k_poa.h"
#include <algorithm>

#ifndef POA_LT
#define POA_LT 0          // experiment (not kept: slower, register pressure): band starts per anchor from an LDS table instead of three divisions per rank
#endif
#ifndef POA_COLD
#define POA_COLD 1       // round 5: cold fields of the job description read from the kernel-argument segment at their use (poa_tile_body)
#endif
#ifndef POA_PHASES
#define POA_PHASES 0     // round 5: the phase-cycle instrumentation (NGSID_POA_PHASES dev aid) is compiled only into dev builds (-DPOA_PHASES=1, tools/micro/build_variant.sh): its pointer, counters and branches sat in the row paths of the product kernel
#endif
#define POA_PHC(J) (POA_PHASES ? (J).phase_cycles : (unsigned long long*)nullptr)
#ifndef POA_REPEAT
#define POA_REPEAT 0      // dev timing builds (tools/micro/build_repeat.sh): 1 / 2 / 3 / 4 = run the prepass / forward pass / traceback / emission twice (same results)
#endif
#define SRC_SLOT 63
#define NONE16 0xFFFFu
#define HR 8           // DP rows kept in the LDS ring
#define RPADL 4        // ring row: RPADL guard cells (0 = minus infinity, see PBIAS) | BW cells | RPADR guard cells, so neighbour reads need no bounds checks
#define RPADR 12
#define DLO_MAX 11     // largest band-start difference to a predecessor a 'near' row may have (<= RPADR - 1)
#define TBR 32         // direction rows per traceback block

#define LDSP __attribute__((address_space(3)))
typedef LDSP uint16_t* l16; typedef LDSP uint8_t* l8; typedef LDSP int32_t* l32; typedef LDSP long long* l64; typedef LDSP unsigned long long* lu64;

struct GG {   // the graph in the workgroup's HBM scratch, every array indexed by the node's TOPOLOGICAL RANK (round 4; oracle/ngsid_oracle_poa_rank.c is the
              // CPU restatement): no node ids, no edge objects.  ONE base pointer + 32-bit byte offsets derived from a few strides.
    uint8_t* base; uint32_t s32, s64, s16, s8, se16, se32, sl;   // bytes of one u32[Vc+1] / u64[Vc+1] / u16[Vc+1] / u8[Vc+1] / u16[Ec] / i32[Ec] / u16[Lmax] array (16-byte multiples)
    // pp: ranks of the tails of the first two in-edges in creation order (p0 | p1 << 16, NONE16 = none); ar: anchor | ring << 16 (ring = rank of the next node
    // aligned to the same column, itself when alone); ww: weights of those two edges (w0 | w1 << 32); cov: count weight
    __device__ __forceinline__ uint32_t& pp(uint32_t i) const { return *(uint32_t*)(base + 4u * i); }
    __device__ __forceinline__ uint16_t& p0(uint32_t i) const { return *(uint16_t*)(base + 4u * i); }
    __device__ __forceinline__ uint16_t& p1(uint32_t i) const { return *(uint16_t*)(base + 4u * i + 2u); }
    __device__ __forceinline__ uint32_t& ar(uint32_t i) const { return *(uint32_t*)(base + s32 + 4u * i); }
    __device__ __forceinline__ uint16_t& anchor(uint32_t i) const { return *(uint16_t*)(base + s32 + 4u * i); }
    __device__ __forceinline__ uint16_t& ring(uint32_t i) const { return *(uint16_t*)(base + s32 + 4u * i + 2u); }
    __device__ __forceinline__ uint32_t& cov(uint32_t i) const { return *(uint32_t*)(base + 2u * s32 + 4u * i); }
    __device__ __forceinline__ unsigned long long& ww(uint32_t i) const { return *(unsigned long long*)(base + 3u * s32 + 8u * i); }
    __device__ __forceinline__ int32_t& w0(uint32_t i) const { return *(int32_t*)(base + 3u * s32 + 8u * i); }
    __device__ __forceinline__ int32_t& w1(uint32_t i) const { return *(int32_t*)(base + 3u * s32 + 8u * i + 4u); }
    __device__ __forceinline__ unsigned long long& ri(uint32_t i) const { return *(unsigned long long*)(base + 3u * s32 + s64 + 8u * i); }   // per-rank row info of the current alignment
    __device__ __forceinline__ long long& sc(uint32_t i) const { return *(long long*)(base + 3u * s32 + 2u * s64 + 8u * i); }                  // heaviest-bundle scores per rank
    __device__ __forceinline__ uint16_t& tmpo(uint32_t i) const { return *(uint16_t*)(base + 3u * s32 + 3u * s64 + 2u * i); }                  // ranks of the bundle path
    __device__ __forceinline__ uint16_t& shiftg(uint32_t i) const { return *(uint16_t*)(base + 3u * s32 + 3u * s64 + s16 + 2u * i); }          // shift[] when it does not fit the LDS
    // cm: letter | 0x80 when the node has more than two in-edges (overflow list); of: 1 = the node has an out-edge; far: 1 = some successor lies more than HR
    // ranks behind (the forward pass keeps an HBM copy of the row).  of / far are written by the lane of the SUCCESSOR, cm by the node's own lane.
    __device__ __forceinline__ uint8_t& cm(uint32_t i) const { return *(base + 3u * s32 + 3u * s64 + 2u * s16 + i); }
    __device__ __forceinline__ uint8_t& of(uint32_t i) const { return *(base + 3u * s32 + 3u * s64 + 2u * s16 + s8 + i); }
    __device__ __forceinline__ uint8_t& far(uint32_t i) const { return *(base + 3u * s32 + 3u * s64 + 2u * s16 + 2u * s8 + i); }
    // third and later in-edges, in creation order
    __device__ __forceinline__ uint16_t& ov_head(uint32_t i) const { return *(uint16_t*)(base + 3u * s32 + 3u * s64 + 2u * s16 + 3u * s8 + 2u * i); }
    __device__ __forceinline__ uint16_t& ov_tail(uint32_t i) const { return *(uint16_t*)(base + 3u * s32 + 3u * s64 + 2u * s16 + 3u * s8 + se16 + 2u * i); }
    __device__ __forceinline__ int32_t& ov_w(uint32_t i) const { return *(int32_t*)(base + 3u * s32 + 3u * s64 + 2u * s16 + 3u * s8 + 2u * se16 + 4u * i); }
    // per sequence position (the alignment being merged) when the LDS cannot hold them: aligned rank, existing / final rank
    __device__ __forceinline__ uint16_t& alnode(uint32_t i) const { return *(uint16_t*)(base + 3u * s32 + 3u * s64 + 2u * s16 + 3u * s8 + 2u * se16 + se32 + 2u * i); }
    __device__ __forceinline__ uint16_t& nodeof(uint32_t i) const { return *(uint16_t*)(base + 3u * s32 + 3u * s64 + 2u * s16 + 3u * s8 + 2u * se16 + se32 + sl + 2u * i); }
};
// LDS working set.  ~10 KB per tile for 750-base reads, so sixteen tiles (four waves per SIMD) are resident per CU.  The hot arrays sit at
// COMPILE-TIME offsets so the row loop spends no SGPRs on them.  Layout (BW = band width), alignment phases | consensus phase:
//   [0, HR*RS*4)            hring   ring of DP rows, RS = RPADL + BW + RPADR ints each (guard cells hold 0)      | epred (2 bytes per rank)
//   [.., + TBR*BW)          dirblk  direction rows: staged by the forward pass, block by block for the traceback      | .. sinkbits
//   [.., + TBR*8)           rblk    row info of the last staged block (hand-over from the forward pass to the traceback)
//   [C1, ..)                sq      (one pad byte in front, BW behind)
// Everything else - the graph, the per-rank row info, the per-position merge arrays, the bundle scores - lives in the L2-resident HBM
// scratch of the workgroup (GG) and is touched by lane-parallel code only.
extern __shared__ __attribute__((aligned(16))) unsigned char poa_smem[];     // dynamic LDS of k_poa_tile (starts at LDS address 0)
#define POA_LDS(T, off) ((T)((LDSP unsigned char*)poa_smem + (off)))
template <int BW>
struct LLT {
    static constexpr unsigned RS = BW + RPADL + RPADR;      // ring row stride (ints)
    static constexpr unsigned HRING = 0, DIRBLK = HR * RS * 4, RBLK = DIRBLK + TBR * BW, C1 = RBLK + TBR * 8;
    static __device__ __forceinline__ l32 hring() { return POA_LDS(l32, HRING); }
    static __device__ __forceinline__ l8 dirblk() { return POA_LDS(l8, DIRBLK); }
    static __device__ __forceinline__ lu64 rblk() { return POA_LDS(lu64, RBLK); }
    static __device__ __forceinline__ l8 sq() { return POA_LDS(l8, C1 + 16); }           // one pad byte in front (sq[-1]), BW behind
    static __device__ __forceinline__ l16 epred() { return POA_LDS(l16, 0); }
    LDSP unsigned int* sinkbits;
};
__host__ __device__ inline size_t poa_al16(size_t b) { return (b + 15) & ~(size_t)15; }
// A per-position u16 array of the alignment being merged (aligned rank / node, chosen node): in LDS when the sequence is short enough for the
// regions that are free at that time (the DP ring during the traceback and the merge, the direction block during the merge), else in the
// workgroup's HBM scratch.  `lds` is wave-uniform: every access is one scalar branch.
struct SeqU16 {
    LDSP uint16_t* l; uint16_t* gm; bool lds;
    __device__ __forceinline__ int get(int i) const { return lds ? (int)l[i] : (int)gm[i]; }
    __device__ __forceinline__ void set(int i, int v) const { if (lds) l[i] = (uint16_t)v; else gm[i] = (uint16_t)v; }
};

// Single-wave workgroup: LDS instructions of one wave execute in issue order, so ordering LDS traffic between lanes only needs the
// compiler not to reorder and the LDS queue to drain - no s_barrier and no wait on outstanding HBM stores.
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
// HBM scratch written by some lanes and read by others of the same wave: drain the stores, then order.
__device__ __forceinline__ void mem_sync() { __threadfence_block(); __syncthreads(); }

// inclusive max-scan over the 64 lanes: the canonical GCN wave scan, six DPP-fused v_max (row_shr 1/2/4/8 inside the 16-lane rows,
// then row_bcast:15 / row_bcast:31 carry the row totals).  In place: lanes without a DPP source keep their value (bound_ctrl off).
// s_nop 1 = the two wait states a DPP read needs after the VALU write of the same register.
__device__ __forceinline__ int wave_incl_max_scan(int v)
{
    asm volatile(
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "+v"(v));
    return v;
}

// the same scan with its input left intact (round 6): the first step is written OUT of place - v_max_i32_dpp dst, v, v row_shr:1 bound_ctrl:1: a lane without a source reads 0
// and takes max(0, v) - so the caller keeps v for the "did the chain win" compare without the register copy the in-place form forces (one v_mov per DP row).  Exact for v >= 0;
// the cell values it is used on are biased by PBIAS (reachable cells ~2^28; an UNREACHABLE cell, whose value and direction nothing ever reads, may come out as 0 instead of a small
// negative number).
__device__ __forceinline__ int wave_incl_max_scan_keep(int v)
{
    int o;
    asm volatile(
        "s_nop 1\n v_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "=&v"(o) : "v"(v));
    return o;
}

// inclusive add-scan over the 64 lanes (same DPP pattern as the max-scan; lanes without a source keep their value)
__device__ __forceinline__ int wave_incl_add_scan(int v)
{
    asm volatile(
        "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
        "s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
        "s_nop 1\n"
        : "+v"(v));
    return v;
}

// Direction bytes (type | slot << 2; slot 0 / 1 = first / second in-edge, SRC_SLOT = virtual source) go to HBM as 4 bits per cell:
// type | code << 2 with code 0, 1, 2 = slot 0, 1, SRC_SLOT.  Rows with more than two in-edges (row flag 2, a fraction of a percent) can hold
// other slots: their byte rows are kept in full next to the packed block; the traceback, which walks the packed blocks as they are, reads
// the byte of such a row from there when the path passes through it.
__device__ __forceinline__ unsigned dir_nib4(unsigned w) { const unsigned t = w & 0x03030303u, sl = (w >> 2) & 0x01010101u, gq = (w >> 7) & 0x01010101u; return t | ((sl & ~gq) << 2) | (gq << 3); }
__device__ __forceinline__ unsigned dir_pack8(unsigned w0, unsigned w1)
{
    const unsigned n0 = dir_nib4(w0), n1 = dir_nib4(w1);
    const unsigned c0 = (n0 | (n0 >> 4)) & 0x00FF00FFu, c1 = (n1 | (n1 >> 4)) & 0x00FF00FFu;
    return ((c0 | (c0 >> 8)) & 0xFFFFu) | (((c1 | (c1 >> 8)) & 0xFFFFu) << 16);
}
__device__ __forceinline__ ngsid_v4u dir_pack32(const ngsid_v4u a, const ngsid_v4u b) { ngsid_v4u o; o.x = dir_pack8(a.x, a.y); o.y = dir_pack8(a.z, a.w); o.z = dir_pack8(b.x, b.y); o.w = dir_pack8(b.z, b.w); return o; }

// phase counters (NGSID_POA_PHASES): 256 slots of 24 counters, one slot per workgroup modulo 256, so that the instrumentation's atomics do not serialise
#define PHS(J) ((J).phase_cycles + (blockIdx.x & 255u) * 24u)
#define PH(J, idx, t0) do { if (POA_PHC(J) && lane == 0) { const unsigned long long t1_ = __builtin_readcyclecounter(); atomicAdd(&PHS(J)[idx], t1_ - (t0)); (t0) = t1_; } } while (0)

struct TS { int V, E, L0, members, nout, capV, capE, nov; unsigned long long cw_sum; };      // nov = entries of the overflow in-edge list

__device__ __forceinline__ int wtof(const PSeq& S, int i) { return S.q ? (int)S.q[i] - 33 : S.uw; }

// band start of a node: centre = trunc((anchor - a0) * len / span) (C division, towards zero), start = centre - BW/2 clamped to [0, len+1-BW].
// The quotient is taken with a precomputed double reciprocal and one exact correction step (operands are < 2^31, so the estimate is
// off by at most one) - a 64-bit integer division costs ~100 instructions per lane and there is one per graph node and alignment.
struct BandMap { int a0; int len; int span; double rcp; int mx; };
__device__ __forceinline__ BandMap band_map(const PSeq& S, int L0, int BW) {
    BandMap m; int a0 = S.a0, a1 = S.a1; if (a1 < a0) { a0 = 0; a1 = L0 - 1; }
    long long span = (long long)a1 - a0 + 1; if (span < 1) span = 1;
    m.a0 = a0; m.len = S.len; m.span = (int)span; m.rcp = 1.0 / (double)span;
    long long mx = (long long)S.len + 1 - BW; if (mx < 0) mx = 0; m.mx = (int)mx;
    return m;
}
__device__ __forceinline__ int band_lo(int anchor, const BandMap& m, int BW) {
    const long long num = (long long)(anchor - m.a0) * (long long)m.len;
    const unsigned long long an = (unsigned long long)(num < 0 ? -num : num);
    long long q = (long long)((double)an * m.rcp);
    long long rem = (long long)an - q * (long long)m.span;
    if (rem < 0) { --q; rem += m.span; }
    if (rem >= (long long)m.span) ++q;
    const long long c = num < 0 ? -q : q;
    long long lo = c - BW / 2;
    if (lo < 0) lo = 0; if (lo > m.mx) lo = m.mx;
    return (int)lo;
}

__device__ __forceinline__ void tile_add_first(const GG& g, const PSeq& S, TS& st, int lane)
{
    for (int i = lane; i < S.len; i += 64) {
        g.cm(i) = S.s[i]; g.ar(i) = (uint32_t)i | ((uint32_t)i << 16); g.cov(i) = S.cw; g.of(i) = (i + 1 < S.len) ? 1 : 0; g.far(i) = 0;
        g.pp(i) = ((i > 0) ? (uint32_t)(i - 1) : (uint32_t)NONE16) | ((uint32_t)NONE16 << 16);
        g.ww(i) = (i > 0) ? (unsigned long long)(unsigned)(wtof(S, i - 1) + wtof(S, i)) : 0ull;
    }
    st.V = S.len; st.E = S.len > 0 ? S.len - 1 : 0; st.L0 = S.len; st.cw_sum += S.cw; st.nov = 0;
    mem_sync();
}

// overflow in-edges (third and later, creation order): index of the first entry at or after `from` whose head is rank r, or -1.  Wave-uniform arguments and
// result; 64 entries per round (the list holds a handful of entries: 0.7 % of the rows have more than two in-edges).
__device__ __forceinline__ int ov_next(const GG& g, int nov, int r, int from, int lane)
{
    for (int x0 = from & ~63; x0 < nov; x0 += 64) {
        const int x = x0 + lane;
        const bool hit = x >= from && x < nov && (int)g.ov_head(x) == r;
        const unsigned long long m = __ballot(hit);
        if (m) return x0 + __builtin_ctzll(m);
    }
    return -1;
}

// one pass of the heaviest-bundle recurrence over ranks [rb, V) in RANK space.  Per 64-rank chunk every lane fetches the first two
// in-edges (predecessor rank, weight) of its rank from HBM into registers; the serial recurrence then runs UNIFORMLY on the whole
// wave: edge data comes through v_readlane, the running scores live in LDS (sc) with the previous rank's score kept in registers,
// so the common case (single predecessor = previous rank) touches no memory on its dependent path.
// completion = branch-completion pass (skip predecessors whose score is -1).  Returns the best rank (uniform) or -1.
__device__ __forceinline__ long long uniform64(long long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}
template <int BW>
__device__ int bundle_pass(const GG& g, const LLT<BW>& w, int V, int nov, int rb, int completion, int lane)
{
    int best = -1; long long best_sv = 0;
    int prev_r = -2; long long prev_sv = 0;                   // score of the rank handled last (register copy of sc[prev_r])
    auto SC = [&](int t) -> long long { return t == prev_r ? prev_sv : uniform64(__builtin_nontemporal_load(&g.sc(t))); };     // L2-served: written by this wave a moment ago
    for (int r0 = rb; r0 < V; r0 += 64) {
        const int r = r0 + lane;
        unsigned tt = NONE16 | (NONE16 << 16), fl = 0; int w0 = 0, w1 = 0;
        if (r < V) {
            tt = g.pp(r); const unsigned long long wv = g.ww(r); w0 = (int)(unsigned)wv; w1 = (int)(unsigned)(wv >> 32);
            if (g.cm(r) & 0x80) fl = 1;
            if (!completion && !g.of(r)) atomicOr((unsigned int*)&w.sinkbits[r >> 5], 1u << (r & 31));
        }
        const int cnt = min(64, V - r0);
        for (int x = 0; x < cnt; ++x) {
            const int rr = r0 + x; long long sv = -1; int pv = NONE16;
            const unsigned tx = __builtin_amdgcn_readlane(tt, x);
            const int ta = tx & 0xffff, tb = tx >> 16;
            const long long wa = (int)__builtin_amdgcn_readlane(w0, x), wb = (int)__builtin_amdgcn_readlane(w1, x);
            if (!__builtin_amdgcn_readlane(fl, x)) {
                bool ha = ta != NONE16, hb = tb != NONE16, ka = false, kb = false; long long sa = 0, sb = 0;
                if (completion) {
                    if (ha) { sa = SC(ta); ka = true; if (sa == -1) ha = false; }
                    if (hb) { sb = SC(tb); kb = true; if (sb == -1) hb = false; }
                }
                if (ha) { sv = wa; pv = ta; }
                if (hb) {
                    bool take = sv < wb;
                    if (!take && sv == wb && ha) {                 // equal weights: the later edge wins unless its tail scores lower
                        if (!ka) { sa = SC(ta); ka = true; }
                        if (!kb) { sb = SC(tb); kb = true; }
                        take = sa <= sb;
                    }
                    if (take) { sv = wb; pv = tb; }
                }
                if (pv != NONE16) sv += (pv == ta) ? (ka ? sa : SC(ta)) : (kb ? sb : SC(tb));
            } else {        // more than two in-edges: the inline pair, then the overflow entries of the rank, in creation order
                long long spv = 0;
                for (int k = 0; k < 2; ++k) {
                    const int t = k ? tb : ta; const long long ww = k ? wb : wa; const long long st_ = SC(t);
                    if (completion && st_ == -1) continue;
                    if (sv < ww || (sv == ww && pv != NONE16 && spv <= st_)) { sv = ww; pv = t; spv = st_; }
                }
                for (int e = ov_next(g, nov, rr, 0, lane); e >= 0; e = ov_next(g, nov, rr, e + 1, lane)) {
                    const int t = __builtin_amdgcn_readfirstlane((int)g.ov_tail(e)); const long long st_ = SC(t);
                    if (completion && st_ == -1) continue;
                    const long long ww = __builtin_amdgcn_readfirstlane(g.ov_w(e));
                    if (sv < ww || (sv == ww && pv != NONE16 && spv <= st_)) { sv = ww; pv = t; spv = st_; }
                }
                if (pv != NONE16) sv += spv;
            }
            if (lane == 0) { g.sc(rr) = sv; w.epred()[rr] = (uint16_t)pv; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the score must have reached L2 before a later rank looks it up (this serial pass is rare)
            prev_r = rr; prev_sv = sv;
            if (best < 0 || best_sv < sv) { best = rr; best_sv = sv; }
        }
    }
    lds_sync();
    return best;
}

// First (complete) pass of the heaviest-bundle recurrence, data-parallel.  The predecessor of a rank is chosen by edge weight alone
// unless two in-edges tie or there are more than two of them ("hard" ranks, a fraction of a percent), so for up to 64 consecutive ranks
// every lane knows its predecessor immediately and the scores sc[r] = w + sc[pred] are path sums: pointer jumping inside the chunk
// (6 rounds of lane shuffles), predecessors before the chunk come from LDS.  A chunk ends in front of the first hard rank whose
// candidates lie inside it; that rank opens the next chunk with all its candidates final.  Same results as bundle_pass(.., 0, 0, ..).
template <int BW>
__device__ int bundle_pass_parallel(const GG& g, const LLT<BW>& w, int V, int nov, int lane)
{
    int best = -1; long long best_sv = 0;
    int r0 = 0;
    while (r0 < V) {
        const int r = r0 + lane; const bool valid = r < V;
        int t0 = NONE16, t1 = NONE16, w0 = 0, w1 = 0; bool many = false;
        if (valid) {
            const uint32_t ppv = g.pp(r); const unsigned long long wv = g.ww(r);
            t0 = ppv & 0xffff; t1 = ppv >> 16; w0 = (int)(unsigned)wv; w1 = (int)(unsigned)(wv >> 32);
            many = (g.cm(r) & 0x80) != 0;
            if (!g.of(r)) atomicOr((unsigned int*)&w.sinkbits[r >> 5], 1u << (r & 31));
        }
        const bool hard = valid && (many || (t1 != NONE16 && w0 == w1));
        // candidates of a hard rank inside the chunk?  (with more than two in-edges: treat as inside - such ranks are rare)
        const bool inside = hard && (many ? true : (t0 >= r0 || t1 >= r0));
        unsigned long long cm_ = __ballot(inside && lane > 0);
        // a rank with many in-edges at lane 0 has all candidates before the chunk by definition
        const int n = min(min(64, V - r0), cm_ ? __ffsll((long long)cm_) - 1 : 64);
        const bool act = lane < n;
        int pv = NONE16; long long val = -1, extv = 0; int ptr = -1;
        if (act) {
            long long wv = -1;
            if (many) {                                   // lane 0 only: the inline pair, then the overflow entries in creation order; all tails are final
                long long spv = 0;
                for (int k = 0; k < 2; ++k) { const int t = k ? t1 : t0; const long long st_ = g.sc(t); const long long ww = k ? w1 : w0; if (wv < ww || (wv == ww && pv != NONE16 && spv <= st_)) { wv = ww; pv = t; spv = st_; } }
                for (int e = 0; e < nov; ++e) if ((int)g.ov_head(e) == r) {
                    const int t = g.ov_tail(e); const long long st_ = g.sc(t); const long long ww = g.ov_w(e);
                    if (wv < ww || (wv == ww && pv != NONE16 && spv <= st_)) { wv = ww; pv = t; spv = st_; }
                }
            } else if (t0 != NONE16) {
                wv = w0; pv = t0;
                if (t1 != NONE16) {
                    bool take = w0 < w1;
                    if (w0 == w1) take = g.sc(t0) <= g.sc(t1);       // hard rank whose candidates are final (both before the chunk)
                    if (take) { wv = w1; pv = t1; }
                }
            }
            val = wv;
            if (pv != NONE16) { if (pv >= r0) ptr = pv - r0; else extv = g.sc(pv); }
        }
        // path sums by pointer jumping (ptr < 0: the chain has left the chunk, extv holds the score it ends on)
#pragma unroll
        for (int round = 0; round < 6; ++round) {
            const int src = ptr >= 0 ? ptr : lane;
            const long long v2 = __shfl(val, src), e2 = __shfl(extv, src); const int p2 = __shfl(ptr, src);
            if (ptr >= 0) { val += v2; extv = e2; ptr = p2; }
        }
        const long long sv = val + extv;
        if (act) { g.sc(r) = sv; w.epred()[r] = (uint16_t)pv; }
        // best of the chunk: largest score, first rank on ties
        long long bv = act ? sv : (long long)0x8000000000000000ll; int bl_ = act ? lane : 64;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const long long ov = __shfl_xor(bv, d); const int ol = __shfl_xor(bl_, d); if (ov > bv || (ov == bv && ol < bl_)) { bv = ov; bl_ = ol; } }
        if (best < 0 || best_sv < bv) { best = r0 + bl_; best_sv = bv; }
        mem_sync();                                       // sc[] of this chunk (HBM) before the next chunk's lookups
        r0 += n;
    }
    return __builtin_amdgcn_readfirstlane(best);
}

// heaviest bundle + branch completion (oracle g_consensus)
template <int BW>
__device__ __forceinline__ void tile_emit(const GG& g, const LLT<BW>& w, const PoaJobSet& J, uint32_t job, TS& st, int lane)
{
    if (st.V == 0 || st.members == 0) return;
    if (st.nout >= J.D) { if (lane == 0 && J.slot_overflow) atomicExch(J.slot_overflow, 1u); return; }   // host retries with more output slots
    const size_t slot = (size_t)job * J.D + st.nout;
    uint8_t* dst = J.out + slot * (size_t)J.Vcap;
    uint32_t* dcov = J.out_cov ? J.out_cov + slot * (size_t)J.Vcap : nullptr;
    const int V = st.V;
    unsigned long long tph = POA_PHC(J) ? __builtin_readcyclecounter() : 0;
    mem_sync();
    for (int x = lane; x < (V + 31) / 32; x += 64) w.sinkbits[x] = 0;
    lds_sync();
    const int nov = st.nov;
    int mx = bundle_pass_parallel(g, w, V, nov, lane);
    unsigned long long tph2 = tph;
    for (int guard = 0; !((w.sinkbits[mx >> 5] >> (mx & 31)) & 1u); ++guard) {
        if (guard > V) { if (lane == 0 && J.slot_overflow) atomicExch(J.slot_overflow + 1, 2u); break; }   // cannot happen: each completion pass starts further down
        const int start = mx;
        // every other in-edge tail of the successors of `start` is taken out (oracle g_consensus).  Lane-parallel over the ranks behind start: a rank is a
        // successor if start is among its in-edge tails (inline pair or overflow entries)
        for (int h0 = start + 1; h0 < V; h0 += 64) {
            const int h = h0 + lane;
            if (h < V) {
                const uint32_t ppv = g.pp(h); const int t0 = ppv & 0xffff, t1 = ppv >> 16; const bool many = (g.cm(h) & 0x80) != 0;
                bool succ = t0 == start || t1 == start;
                if (many && !succ) for (int e = 0; e < nov; ++e) if ((int)g.ov_head(e) == h && (int)g.ov_tail(e) == start) succ = true;
                if (succ) {
                    if (t0 != NONE16 && t0 != start) g.sc(t0) = -1;
                    if (t1 != NONE16 && t1 != start) g.sc(t1) = -1;
                    if (many) for (int e = 0; e < nov; ++e) if ((int)g.ov_head(e) == h && (int)g.ov_tail(e) != start) g.sc(g.ov_tail(e)) = -1;
                }
            }
        }
        mem_sync();
        const int m2 = bundle_pass(g, w, V, nov, start + 1, 1, lane);
        if (m2 < 0) break;
        mx = m2;
    }
    // backtrack: lane 0 lists the ranks of the path (HBM scratch), then all lanes translate rank -> letter / coverage
    int n = 0;
    {   // most of the path is "predecessor = previous rank": lane k inspects rank r-k and the wave takes the whole leading run at once.
        // One walk; ranks are written from the END of the scratch array, the path then starts at tmpo[V - n].
        int i = V, r = mx;
        while (r != NONE16) {
            const int rk = r - lane;
            const int pk = rk >= 0 ? (int)w.epred()[rk] : NONE16;
            const unsigned long long gm = __ballot(rk >= 1 && pk == rk - 1) & 0x7fffffffffffffffull;     // at most 63 chained steps per round
            const int run = __builtin_ctzll(~gm);          // ranks r .. r-run+1 step to their previous rank; rank r-run is reached and decides what follows
            if (lane <= run) g.tmpo(i - 1 - lane) = (uint16_t)rk;
            i -= run + 1;
            r = __builtin_amdgcn_readlane(pk, run);        // predecessor of rank r-run (NONE16 ends the path)
        }
        n = V - i;
    }
    n = __builtin_amdgcn_readfirstlane(n);
    const int poff = V - n;
    mem_sync();
    PH(J, 15, tph2);
    const int trim_tiles = (J.job_final && J.job_final[job]) ? 0 : J.trim_tiles;      // (trim 3: the tile that ends a unit is not trimmed)
    const bool trim = trim_tiles && dcov && n > 0;        // oracle EMIT: coverage-trim the tile consensus ends
    const uint32_t thr = (uint32_t)(st.cw_sum / 2);
    int tb = 0x7fffffff, te = -1;                            // first / last position whose column carries at least half of the merged weight
    for (int i = lane; i < n; i += 64) {
        const int v = g.tmpo(poff + i); dst[i] = g.cm(v) & 0x7f;
        if (dcov) { uint32_t c = g.cov(v); for (int u = g.ring(v); u != v; u = g.ring(u)) c += g.cov(u); dcov[i] = c; if (c >= thr) { tb = min(tb, i); te = max(te, i); } }
    }
    int span_b = 0, span_e = n - 1;              // consensus positions whose nodes give the span (a0, a1) of the output in the first sequence's coordinates
    if (trim) {
        int b = tb, e = te;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { b = min(b, __shfl_xor(b, d)); e = max(e, __shfl_xor(e, d)); }
        if (b < e && b != 0x7fffffff) {
            span_b = b; span_e = e;
            // upper levels (trim_tiles & 2, members = weighted tile consensuses): between the kept ends every base whose column carries less than a THIRD of the
            // merged weight goes as well (oracle EMIT: the heaviest bundle maximises the SUM of the edge weights of a path, so a k-base insertion of the weight w
            // beats the direct edge W of the rest when (k + 1) w > W - a third for one base, a seventh for five).  Level-0 tiles keep spoa's / racon's behaviour.
            // Ordered in-place compaction, 64 positions per round: a round's stores land below the next round's loads (an output index never exceeds the input
            // index), and a round's own loads have returned before its stores issue (they carry the loaded values): one drain in front of the loop suffices.
            const uint32_t thr3 = (trim_tiles & 2) ? (uint32_t)(st.cw_sum / 3) : 0u;
            const int m2 = e - b + 1;
            if (thr3 == 0u && b == 0) n = m2;                 // nothing moves
            else {
                int kept = 0;
                mem_sync();
                for (int c0 = 0; c0 < m2; c0 += 64) {
                    const int x = c0 + lane; uint8_t ch = 0; uint32_t cv = 0;
                    if (x < m2) { ch = dst[b + x]; cv = dcov[b + x]; }
                    const bool keep = x < m2 && cv >= thr3;
                    const unsigned long long km = __ballot(keep);
                    if (keep) { const int o = kept + __popcll(km & ((1ull << lane) - 1ull)); dst[o] = ch; dcov[o] = cv; }
                    kept += __popcll(km);
                }
                n = kept;
            }
        }
    }
    if (lane == 0) {
        J.out_len[slot] = n; J.out_cw[slot] = st.cw_sum;
        if (J.out_span) { J.out_span[2 * slot] = n > 0 ? (int32_t)g.anchor(g.tmpo(poff + span_b)) : 0; J.out_span[2 * slot + 1] = n > 0 ? (int32_t)g.anchor(g.tmpo(poff + span_e)) : -1; }
    }
    if (POA_PHC(J) && lane == 0) atomicAdd(&PHS(J)[12], (unsigned long long)n);
    mem_sync();
    PH(J, 4, tph);
    st.nout += 1;
}

// Forward DP over the ranks of the graph (one row per node, topological order).  LOCAL is a template constant so the clamp / best-cell
// bookkeeping of the other modes costs nothing.  Two kinds of rows:
//   * chain rows (flag 16: the only predecessor is the previous row and the band start moves by 0 or 1; ~85 % of all rows): inputs come
//     from the previous row's registers and ONE DPP lane shift.  Runs of chain rows execute in an inner loop that contains no HBM
//     instruction at all (so the compiler's waitcnt model has nothing to wait for) and no synchronisation;
//   * everything else: predecessor rows from the LDS ring (HBM copy when further than HR rows back), slot bookkeeping for the traceback.
// Direction bytes are staged in the LDS block the traceback will use later (aligned blocks of TBR rows) and flushed to HBM with 16-byte
// stores when a block fills; the final (partial) block stays in LDS for the traceback.
// Out: the best end cell as (value, rank << 8 | band column), ties -> lowest rank, then lowest column.
template <int CPL>
__device__ __forceinline__ void poa_row_tail_store(l32 ringrow, LDSP uint8_t* dst, const int (&h)[CPL], unsigned dpack)
{
#pragma unroll
    for (int c = 0; c < CPL; ++c) ringrow[c] = h[c];
    if (CPL == 1) *dst = (uint8_t)dpack; else if (CPL == 2) *(LDSP uint16_t*)dst = (uint16_t)dpack; else *(LDSP unsigned int*)dst = dpack;
}

// Stored cell values.  The cell of band column j is kept as S = H - j*g + PBIAS (H = the score the oracle computes):
//   * the column skew j*g turns the in-row gap chain H[j] = max(X[j], H[j-1] + g) into a plain prefix maximum, a vertical move into S_up + g, a
//     diagonal move into S_diag + (s - g) and the free-start terms of the semi-global / source rows into constants: the row loop carries no
//     per-lane column offsets (only the local mode still needs its floor 0 = PBIAS - j*g);
//   * the bias makes "minus infinity" = anything near 0, which is what the guard cells hold and what a DPP lane shift leaves in lanes without a
//     source: reachable cells are PBIAS +- 2^20, unreachable ones stay below 2^20 in magnitude (they are not normalised: every comparison that
//     decides a reachable cell is unaffected, and their own direction bytes are never read by the traceback).
// All differences between candidates of one cell are the same as in H space, so every maximum and every tie-break is the oracle's.
// Band columns past the end of the sequence (only when L+1 < BW) see the 0xFF padding and can never beat a real cell.
#define PBIAS (1 << 28)
#ifndef POA_SCAN_KEEP
#define POA_SCAN_KEEP 1
#endif
template <int CPL, bool LOCAL>
__device__ __forceinline__ unsigned poa_row_finish(const int (&X)[CPL], const int (&Dd)[CPL], int floor0, int gp, int (&hout)[CPL])
{
    if constexpr (CPL == 1) {
        // One cell per lane: in S space the in-row gap chain IS the inclusive prefix maximum, so the scan result is the cell's value; the cell was reached through
        // the chain iff that value differs from its own candidate, and (local mode) it sits on the floor iff the value equals the floor (the floor grows with the
        // column, so the prefix maximum of the floors is the lane's own).  Same values and directions as the general form below, two instructions less per row.
        const int xf = LOCAL ? max(X[0], floor0) : X[0];
#if POA_SCAN_KEEP
        const int incl = LOCAL ? wave_incl_max_scan(xf) : wave_incl_max_scan_keep(xf);      // (local mode: xf is a temporary already)
#else
        const int incl = wave_incl_max_scan(xf);
#endif
        int dd = Dd[0];
        if (incl != X[0]) dd = 2;
        if (LOCAL && incl == floor0) dd = 3;
        hout[0] = incl;
        return (unsigned)(dd & 0xff);
    }
    int exl[CPL]; int run = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int xf = LOCAL ? max(X[c], floor0 - c * gp) : X[c];
        exl[c] = run; run = c ? max(run, xf) : xf;
    }
    const int incl = wave_incl_max_scan(run);
    const int excl_lane = __builtin_amdgcn_update_dpp(0, incl, 0x138, 0xf, 0xf, true);      // wave_shr:1, lane 0 gets 0 = "minus infinity"
    unsigned dpack = 0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int ex = c ? max(excl_lane, exl[c]) : excl_lane;       // best value reachable through the in-row gap chain
        int val = X[c], dd = Dd[c];
        if (ex > val) { val = ex; dd = 2; }
        if (LOCAL && val <= floor0 - c * gp) { val = floor0 - c * gp; dd = 3; }
        hout[c] = val; dpack |= (unsigned)(dd & 0xff) << (8 * c);
    }
    return dpack;
}

template <int CPL, int MODE>
__device__ __forceinline__ void poa_forward(const GG& g, const LLT<64 * CPL>& w, int32_t* Hg, uint8_t* Dg, uint8_t* Dfull, const PSeq& S, int V, int nov, int gp_, int sm_, int sn_, int lane, int& bestv_out, int& bestpk_out, int& nslow_out)
{
    constexpr int BW = 64 * CPL;
    int nslow = 0;
    constexpr bool LOCAL = MODE == NGSID_POA_LOCAL, semi = MODE == NGSID_POA_SEMI;
    const int L = __builtin_amdgcn_readfirstlane(S.len);
    int gp = gp_, sm = sm_ - gp_, sn = sn_ - gp_;         // sm / sn: diagonal increments in S space (s - g)
    const int lane_jg = lane * CPL * gp_;
    asm volatile("" : "+v"(gp), "+v"(sm), "+v"(sn));      // keep the three score constants in VGPRs (SGPRs are the scarce resource of this kernel)
    int bestv = PBIAS / 2, bestpk = 0x7fffffff; // !LOCAL (S space; every candidate sits in column L, so the order is the order of H)
    unsigned bkey[CPL];                         // LOCAL: (H << 16) | (0xFFFF - rank) per owned band column; H is >= 0 and < 2^16 (host check)
    int hprev[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { hprev[c] = 0; bkey[c] = 0; }
    constexpr int RS = BW + RPADL + RPADR;
    const l32 ring0 = w.hring() + RPADL + lane * CPL;
    const l8 stage0 = w.dirblk() + lane * CPL;
    const l8 sq0 = w.sq() + lane * CPL - 1;
    // row info of ranks [r & ~63, +64) in lane order (one v_readlane pair per row, no LDS on the row's critical path); the next 64 are
    // prefetched from HBM a whole chunk ahead
    unsigned clo, chi, nlo, nhi;
    { const unsigned long long a = lane < V ? g.ri(lane) : 0ull, b = 64 + lane < V ? g.ri(64 + lane) : 0ull; clo = (unsigned)a; chi = (unsigned)(a >> 32); nlo = (unsigned)b; nhi = (unsigned)(b >> 32); }
    int r = 0;
    while (r < V) {
        unsigned rlo = __builtin_amdgcn_readlane(clo, r & 63), rhi = __builtin_amdgcn_readlane(chi, r & 63);
        int rfl = rhi >> 24;
        if (rfl & 128) {
            // ---- tight run (see the prepass): n chain rows, band start + 1 per row -> diagonal = own register, up = lane+1's (one DPP);
            //      nothing to decode per row but the node letter
            int n = rhi & 0xff, l0 = rlo & 0xffff;
            int q[CPL], qn[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) qn[c] = sq0[l0 + c];
#pragma unroll
            for (int c = 0; c < CPL; ++c) asm volatile("" : "+v"(qn[c]));      // consumed here: the loop header then carries no pending LDS read, so the
                                                                              // loop waits for its own prefetch only (lgkmcnt(2)), never for the row stores
            int cvn = (__builtin_amdgcn_readlane(chi, r & 63) >> 16) & 0xff;
            // one row of the run: qa = this row's letters (already in registers), qb receives the next row's.  Two rows per loop iteration (round 4): the letter
            // registers swap roles instead of being copied (q = qn) and the row's candidate needs no copy in front of the scan at the loop's back edge
            auto tight_row = [&](int (&qa)[CPL], int (&qb)[CPL]) {
                const int cv = cvn;
                cvn = (__builtin_amdgcn_readlane(chi, (r + 1) & 63) >> 16) & 0xff;     // next row's letter (an unused lane read at the end of a chunk)
#pragma unroll
                for (int c = 0; c < CPL; ++c) qb[c] = sq0[l0 + 1 + c];      // letters of the next row: their LDS latency hides behind this row (sq is padded)
                const int rt = __builtin_amdgcn_update_dpp(0, hprev[0], 0x130, 0xf, 0xf, true);             // lane+1's first column
                int X[CPL], Dd[CPL];
                int floor0 = 0;
                if (LOCAL) { int l0g = l0 * gp_; asm volatile("" : "+s"(l0g)); floor0 = PBIAS - l0g - lane_jg; }
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int sc = qa[c] == cv ? sm : sn;
                    const int xu = (c + 1 < CPL ? hprev[c + 1] : rt) + gp; int xd = hprev[c] + sc, ds = 0;
                    if (semi) { const int sv = sc + PBIAS; if (l0 + lane * CPL + c >= 1 && sv > xd) { xd = sv; ds = SRC_SLOT << 2; } }
                    X[c] = max(xd, xu); Dd[c] = xu > xd ? 1 : ds;
                }
                const unsigned dpack = poa_row_finish<CPL, LOCAL>(X, Dd, floor0, gp, hprev);
                poa_row_tail_store<CPL>(ring0 + (r & (HR - 1)) * RS, stage0 + (r & (TBR - 1)) * BW, hprev, dpack);
                if (LOCAL) {
                    const unsigned rk = 0xFFFFu - (unsigned)r;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) bkey[c] = max(bkey[c], ((unsigned)(hprev[c] - (floor0 - c * gp)) << 16) | rk);
                }
                ++r; ++l0;
            };
            for (; n >= 2; n -= 2) { tight_row(qn, q); tight_row(q, qn); }
            if (n) tight_row(qn, q);
            rfl = 0;                                                           // nothing pending for the rows just done
        } else if (rfl & 16) {
            // ---- run of chain rows: registers + one DPP per row, LDS only for letters / ring / staged directions
            for (;;) {
                const int l0 = rlo & 0xffff; const int cv = (rhi >> 16) & 0xff;
                int q[CPL];
#pragma unroll
                for (int c = 0; c < CPL; ++c) q[c] = sq0[l0 + c];          // sq is padded: no bounds branches
                // Band start moved by one: diag = same register, up = next column; else diag = previous column, up = same.
                // Lane 0 / 63 get 0 from the DPP (no source lane), which is the out-of-band value; column 0 never has a diagonal.
                // the two candidates are formed inside each band-shift case, so that the lane shift folds into the add (v_add_u32_dpp, as in the tight loop) and the cases
                // need no register copies to meet (round 4)
                int sc[CPL], xu[CPL], xdv[CPL];
#pragma unroll
                for (int c = 0; c < CPL; ++c) sc[c] = q[c] == cv ? sm : sn;
                if (rfl & 32) {
                    const int rt = __builtin_amdgcn_update_dpp(0, hprev[0], 0x130, 0xf, 0xf, true);         // lane+1's first column
#pragma unroll
                    for (int c = 0; c < CPL; ++c) { xu[c] = (c + 1 < CPL ? hprev[c + 1] : rt) + gp; xdv[c] = hprev[c] + sc[c]; }
                } else {
                    const int lf = __builtin_amdgcn_update_dpp(0, hprev[CPL - 1], 0x138, 0xf, 0xf, true);   // lane-1's last column
#pragma unroll
                    for (int c = 0; c < CPL; ++c) { xu[c] = hprev[c] + gp; xdv[c] = (c ? hprev[c - 1] : lf) + sc[c]; }
                }
                int X[CPL], Dd[CPL];
                int floor0 = 0;
                if (LOCAL) { int l0g = l0 * gp_; asm volatile("" : "+s"(l0g)); floor0 = PBIAS - l0g - lane_jg; }      // scalar product (no per-lane v_mul_lo_u32)
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    int xd = xdv[c], ds = 0;
                    if (semi) { const int sv = sc[c] + PBIAS; if (l0 + lane * CPL + c >= 1 && sv > xd) { xd = sv; ds = SRC_SLOT << 2; } }   // free start in the graph
                    X[c] = max(xd, xu[c]); Dd[c] = xu[c] > xd ? 1 : ds;
                }
                const unsigned dpack = poa_row_finish<CPL, LOCAL>(X, Dd, floor0, gp, hprev);
                poa_row_tail_store<CPL>(ring0 + (r & (HR - 1)) * RS, stage0 + (r & (TBR - 1)) * BW, hprev, dpack);
                if (LOCAL) {
                    const unsigned rk = 0xFFFFu - (unsigned)r;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) bkey[c] = max(bkey[c], ((unsigned)(hprev[c] - (floor0 - c * gp)) << 16) | rk);
                } else if ((semi || (rfl & 4)) && (unsigned)(L - l0) < (unsigned)BW) {     // end cells: last column, on sinks (any row in semi-global mode)
#pragma unroll
                    for (int c = 0; c < CPL; ++c) if (l0 + lane * CPL + c == L && hprev[c] > bestv) { bestv = hprev[c]; bestpk = (r << 8) | (lane * CPL + c); }
                }
                ++r;
                if ((rfl & 8) || (r & (TBR - 1)) == 0 || r >= V) break;       // HBM copy / block flush / chunk switch / end: handled below
                rlo = __builtin_amdgcn_readlane(clo, r & 63); rhi = __builtin_amdgcn_readlane(chi, r & 63); rfl = rhi >> 24;
                if (!(rfl & 16)) { rfl = 0; break; }                           // next row is not a chain row (nothing pending for the row just done)
            }
        } else if (rfl & 64) {
            // ---- near row: one or two predecessors, both still in the LDS ring; all reads issue at once, guard cells replace bounds checks
            const int l0 = rlo & 0xffff; const int cv = (rhi >> 16) & 0xff;
            const int dist0 = (rlo >> 16) & 0xff, dist1 = rlo >> 24, dlo0 = rhi & 0xff, dlo1 = (rhi >> 8) & 0xff;
            asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();     // ring rows of earlier iterations: LDS executes a wave's instructions in order
            const l32 H0 = ring0 + ((r - dist0) & (HR - 1)) * RS + (dlo0 - 1);
            int h0[CPL + 1], h1[CPL + 1];
#pragma unroll
            for (int c = 0; c <= CPL; ++c) h0[c] = H0[c];
            if (dist1) {
                const l32 H1 = ring0 + ((r - dist1) & (HR - 1)) * RS + (dlo1 - 1);
#pragma unroll
                for (int c = 0; c <= CPL; ++c) h1[c] = H1[c];
            } else {
#pragma unroll
                for (int c = 0; c <= CPL; ++c) h1[c] = 0;
            }
            int X[CPL], Dd[CPL];
            int floor0 = 0;
            if (LOCAL) { int l0g = l0 * gp_; asm volatile("" : "+s"(l0g)); floor0 = PBIAS - l0g - lane_jg; }
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int sc = ((int)sq0[l0 + c] == cv) ? sm : sn;
                // first predecessor wins ties (the oracle walks the edge list in order and replaces on strictly greater)
                const int xu = max(h0[c + 1], h1[c + 1]) + gp, us = h1[c + 1] > h0[c + 1] ? 1 : 0;
                int xd = max(h0[c], h1[c]) + sc, ds = h1[c] > h0[c] ? 1 : 0;
                if (semi) { const int sv = sc + PBIAS; if (l0 + lane * CPL + c >= 1 && sv > xd) { xd = sv; ds = SRC_SLOT; } }       // free start in the graph
                if (xd >= xu) { X[c] = xd; Dd[c] = 0 | (ds << 2); } else { X[c] = xu; Dd[c] = 1 | (us << 2); }
            }
            const unsigned dpack = poa_row_finish<CPL, LOCAL>(X, Dd, floor0, gp, hprev);
            poa_row_tail_store<CPL>(ring0 + (r & (HR - 1)) * RS, stage0 + (r & (TBR - 1)) * BW, hprev, dpack);
            if (LOCAL) {
                const unsigned rk = 0xFFFFu - (unsigned)r;
#pragma unroll
                for (int c = 0; c < CPL; ++c) bkey[c] = max(bkey[c], ((unsigned)(hprev[c] - (floor0 - c * gp)) << 16) | rk);
            } else if ((semi || (rfl & 4)) && (unsigned)(L - l0) < (unsigned)BW) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) if (l0 + lane * CPL + c == L && hprev[c] > bestv) { bestv = hprev[c]; bestpk = (r << 8) | (lane * CPL + c); }
            }
            ++r;
        } else {
            // ---- generic row (no predecessor, semi-global mode, far or many predecessors): walk the in-edge list in HBM
            ++nslow;
            const int l0 = rlo & 0xffff; const int cv = (rhi >> 16) & 0xff;
            const int jb = l0 + lane * CPL;
            asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
            const bool nopred = (rfl & 1) != 0;
            const bool use_src = nopred || semi;
            int scj[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) scj[c] = ((int)sq0[l0 + c] == cv) ? sm : sn;
            int Xd[CPL], Dslot[CPL], Xu[CPL], Uslot[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) { Xd[c] = 0; Xu[c] = 0; Dslot[c] = 0; Uslot[c] = 0; }
            // in-edges in creation order: the inline pair, then the overflow entries of this rank (slot = position in that order, as the traceback decodes it)
            const uint32_t ppv = nopred ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readfirstlane((int)g.pp(r));
            const bool many_r = !nopred && (__builtin_amdgcn_readfirstlane((int)g.cm(r)) & 0x80);
            int ovx = -1;
            for (int slot = 0;; ++slot) {
                int pr;
                if (slot == 0) { pr = ppv & 0xffff; if (pr == NONE16) break; }
                else if (slot == 1) { pr = ppv >> 16; if (pr == NONE16) break; }
                else { if (!many_r) break; ovx = ov_next(g, nov, r, ovx + 1, lane); if (ovx < 0) break; pr = __builtin_amdgcn_readfirstlane((int)g.ov_tail(ovx)); }
                const int plo = (int)(__builtin_amdgcn_readfirstlane((unsigned)g.ri(pr)) & 0xffff);
                const int pc0 = jb - plo;
                int hp[CPL + 1];
                if ((r - pr) <= HR) {                       // LDS ring
                    const l32 Hp = w.hring() + (size_t)(pr & (HR - 1)) * RS + RPADL;
#pragma unroll
                    for (int c = 0; c <= CPL; ++c) { const int pc = pc0 - 1 + c; hp[c] = (pc >= 0 && pc < BW) ? Hp[pc] : 0; }
                } else {                                    // far predecessor: HBM copy of the row (flag 8 made the producer store it)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int32_t* Hq = Hg + (size_t)pr * BW;
#pragma unroll
                    for (int c = 0; c <= CPL; ++c) { const int pc = pc0 - 1 + c; hp[c] = (pc >= 0 && pc < BW) ? __builtin_nontemporal_load(Hq + pc) : 0; }
                }
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int j = jb + c;
                    { const int hv = hp[c + 1]; if (hv > PBIAS / 2 && hv + gp > Xu[c]) { Xu[c] = hv + gp; Uslot[c] = slot; } }
                    if (j >= 1) { const int hv = hp[c]; if (hv > PBIAS / 2 && hv + scj[c] > Xd[c]) { Xd[c] = hv + scj[c]; Dslot[c] = slot; } }
                }
            }
            int X[CPL], Dd[CPL];
            const int floor0 = PBIAS - l0 * gp_ - lane_jg;       // local mode: S of the score 0 in the lane's first column
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = jb + c;
                // virtual source: score 0 (local) or j*g (global) in column j, i.e. S = floor (local) / PBIAS (global)
                if (use_src && j >= 1) { const int sv = (LOCAL ? floor0 - c * gp + gp : PBIAS) + scj[c]; if (sv > Xd[c]) { Xd[c] = sv; Dslot[c] = SRC_SLOT; } }
                if (nopred && !semi) { const int sv = (LOCAL ? floor0 - c * gp : PBIAS) + gp; if (sv > Xu[c]) { Xu[c] = sv; Uslot[c] = SRC_SLOT; } }
                if (Xd[c] >= Xu[c]) { X[c] = Xd[c]; Dd[c] = 0 | (Dslot[c] << 2); } else { X[c] = Xu[c]; Dd[c] = 1 | (Uslot[c] << 2); }
            }
            const unsigned dpack = poa_row_finish<CPL, LOCAL>(X, Dd, floor0, gp, hprev);
            poa_row_tail_store<CPL>(ring0 + (r & (HR - 1)) * RS, stage0 + (r & (TBR - 1)) * BW, hprev, dpack);
            if (LOCAL) {
                const unsigned rk = 0xFFFFu - (unsigned)r;
#pragma unroll
                for (int c = 0; c < CPL; ++c) bkey[c] = max(bkey[c], ((unsigned)(hprev[c] - (floor0 - c * gp)) << 16) | rk);
            } else if (semi || (rfl & 4)) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) if (jb + c == L && hprev[c] > bestv) { bestv = hprev[c]; bestpk = (r << 8) | (lane * CPL + c); }
            }
            ++r;
        }
        // ---- after the row(s): HBM copy of row r-1 where a far successor will ask for it; flush a completed block of direction rows
        if (rfl & 8) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) Hg[(size_t)(r - 1) * BW + lane * CPL + c] = hprev[c];
        }
        if ((r & (TBR - 1)) == 0) {
            asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
            uint8_t* dstb = Dg + (size_t)(r - TBR) * (BW / 2);            // packed: 4 bits per cell
#pragma unroll
            for (int x = 0; x < TBR * BW / 32 / 64; ++x) {
                const int q = lane + 64 * x;
                const ngsid_v4u a = *(LDSP ngsid_v4u*)(w.dirblk() + q * 32), b = *(LDSP ngsid_v4u*)(w.dirblk() + q * 32 + 16);
                *(ngsid_v4u*)(dstb + q * 16) = dir_pack32(a, b);
            }
            // rows with more than two in-edges keep their byte rows (row info of the block's ranks is still in this chunk's registers)
            unsigned long long irr = __ballot(((chi >> 24) & 2u) != 0) & (0xFFFFFFFFull << ((r - TBR) & 63));
            while (irr) {
                const int bq = __builtin_ctzll(irr); irr &= irr - 1;
                const int row = ((r - 1) & ~63) + bq;
                if (lane < BW / 16) *(ngsid_v4u*)(Dfull + (size_t)row * BW + lane * 16) = *(LDSP ngsid_v4u*)(w.dirblk() + (row & (TBR - 1)) * BW + lane * 16);
            }
            asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
        }
        if ((r & 63) == 0 && r < V) {          // next chunk of row info; prefetch the one after
            clo = nlo; chi = nhi;
            const unsigned long long b = r + 64 + lane < V ? g.ri(r + 64 + lane) : 0ull; nlo = (unsigned)b; nhi = (unsigned)(b >> 32);
        }
    }
    {   // the traceback starts in the last block of direction rows, which is still staged in LDS: give it the row info of those rows and
        // pack the block in place (4 bits per cell, the layout every other block comes back from HBM in)
        const int cb = (V - 1) & ~63, blk = (V - 1) & ~(TBR - 1), x = cb + lane;
        if (x >= blk && x < V) w.rblk()[x - blk] = (unsigned long long)clo | ((unsigned long long)chi << 32);
        asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
        constexpr int NP = TBR * BW / 32 / 64;
        ngsid_v4u pk[NP];
#pragma unroll
        for (int xq = 0; xq < NP; ++xq) {
            const int q = lane + 64 * xq;
            const ngsid_v4u a = *(LDSP ngsid_v4u*)(w.dirblk() + q * 32), b = *(LDSP ngsid_v4u*)(w.dirblk() + q * 32 + 16);
            pk[xq] = dir_pack32(a, b);
        }
        if (V & (TBR - 1)) {         // a partial block was never flushed: its rows with more than two in-edges go to the byte rows now
            unsigned long long irr = __ballot(((chi >> 24) & 2u) != 0) & (0xFFFFFFFFull << (blk & 63));
            while (irr) {
                const int bq = __builtin_ctzll(irr); irr &= irr - 1;
                const int row = cb + bq;
                if (lane < BW / 16) *(ngsid_v4u*)(Dfull + (size_t)row * BW + lane * 16) = *(LDSP ngsid_v4u*)(w.dirblk() + (row & (TBR - 1)) * BW + lane * 16);
            }
        }
        lds_sync();
#pragma unroll
        for (int xq = 0; xq < NP; ++xq) *(LDSP ngsid_v4u*)(w.dirblk() + (lane + 64 * xq) * 16) = pk[xq];
#if POA_REPEAT == 3
#pragma unroll
        for (int xq = 0; xq < NP; ++xq) *(ngsid_v4u*)(Dg + (size_t)blk * (BW / 2) + (lane + 64 * xq) * 16) = pk[xq];
#endif
    }
    if (LOCAL) {
        unsigned k = bkey[0]; int cc = 0;
#pragma unroll
        for (int c = 1; c < CPL; ++c) if (bkey[c] > k) { k = bkey[c]; cc = c; }
        bestv = (int)(k >> 16); bestpk = (int)(((0xFFFFu - (k & 0xFFFFu)) << 8) | (unsigned)(lane * CPL + cc));
    }
    else bestv = bestv - PBIAS + L * gp_;       // back to H (lanes without a candidate stay far below every real score)
    bestv_out = bestv; bestpk_out = bestpk; nslow_out = nslow;
}

#ifndef POA_FWD_CALL
#define POA_FWD_CALL 0      // experiment: the forward pass as a real (non-inlined) function with a register allocation of its own
#endif
#if POA_FWD_CALL
struct FwdOut { int bestv, bestpk, nslow; };
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) { return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v); }
template <int CPL, int MODE>
__device__ __attribute__((noinline)) FwdOut poa_forward_call(unsigned long long gbase, uint32_t s32, uint32_t s64, uint32_t s16, uint32_t s8, uint32_t se16, uint32_t se32, uint32_t sl,
                                                              unsigned long long hg, unsigned long long dg, unsigned long long dfull, int slen, int V, int nov, int gp, int sm, int sn, int lane)
{
    // arguments arrive in VGPRs: make the wave-uniform ones scalar again (once per alignment)
    GG g; g.base = (uint8_t*)uni64(gbase);
    g.s32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s32); g.s64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s64); g.s16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s16); g.s8 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s8);
    g.se16 = (uint32_t)__builtin_amdgcn_readfirstlane((int)se16); g.se32 = (uint32_t)__builtin_amdgcn_readfirstlane((int)se32); g.sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl);
    LLT<64 * CPL> w; w.sinkbits = nullptr;
    PSeq S; S.s = nullptr; S.q = nullptr; S.len = __builtin_amdgcn_readfirstlane(slen); S.uw = 0; S.cw = 0; S.mode = MODE; S.a0 = 0; S.a1 = -1;
    FwdOut o;
    poa_forward<CPL, MODE>(g, w, (int32_t*)uni64(hg), (uint8_t*)uni64(dg), (uint8_t*)uni64(dfull), S, __builtin_amdgcn_readfirstlane(V), __builtin_amdgcn_readfirstlane(nov),
                           __builtin_amdgcn_readfirstlane(gp), __builtin_amdgcn_readfirstlane(sm), __builtin_amdgcn_readfirstlane(sn), lane, o.bestv, o.bestpk, o.nslow);
    return o;
}
#endif

// align S to the graph and merge it.  returns 0 = dropped (no valid end cell), 1 = added, 2 = does not fit
template <int CPL>
__device__ __forceinline__ int tile_align_add(const GG& g, const LLT<64 * CPL>& w, int32_t* Hg, uint8_t* Dg, uint8_t* Dfull, const PoaJobSet& J, const PSeq& S, TS& st, int lane, int& edge_out)
{
    constexpr int BW = 64 * CPL;
    // wave-uniform by construction; tell the compiler so the row loops get scalar control flow
    const int L = __builtin_amdgcn_readfirstlane(S.len), mode = __builtin_amdgcn_readfirstlane(S.mode), gp = __builtin_amdgcn_readfirstlane(J.g), V = __builtin_amdgcn_readfirstlane(st.V);
    unsigned long long tph = POA_PHC(J) ? __builtin_readcyclecounter() : 0;
    // ---------- per-rank row info, built lane-parallel so that the serial row loop reads ONE 8-byte LDS word per row:
    //   lo:16 | dist0:8 | dist1:8 | dlo0:8 | dlo1:8 | letter:8 | flags:8
    //   dist = rank distance to the first / second predecessor (0 = none), dlo = band start of this row minus that of the predecessor
    //   flags: 1 no predecessor, 2 irregular (more than two predecessors, or a distance / band shift that does not fit): walk the
    //          edge list in HBM, 4 sink, 8 keep an HBM copy (a successor is > HR rows away),
    //          16 chain row (single predecessor = previous row, band shift 0/1), 32 = that band shift,
    //          64 near row (one or two predecessors, all within the LDS ring, band shifts 0..DLO_MAX),
    //          128 first row of a tight run of chain rows (its length replaces dlo0; see the last pass)
    const BandMap bm = band_map(S, st.L0, BW);
    // One STREAMING pass, two 64-rank chunks per iteration: the graph is stored in rank order, so a rank's record (tails of its first two in-edges, anchor,
    // letter, flags) is one coalesced load per array; the only dependent loads are the anchors of the two predecessors (their band starts are recomputed
    // from them).  far[] (a successor more than HR ranks behind: the row needs an HBM copy) is a graph property kept by the merge, so the rows of a tight
    // run are known here and no second pass is needed.
    const bool seq_lds = (unsigned)L * 2u <= (unsigned)(HR * (BW + RPADL + RPADR) * 4) && (unsigned)L * 2u <= (unsigned)(TBR * BW) && (unsigned)((L + 63) / 64) * 12u <= (unsigned)(TBR * 8);
    const SeqU16 alnode = { POA_LDS(l16, LLT<BW>::HRING), &g.alnode(0), seq_lds }, nodeof = { POA_LDS(l16, LLT<BW>::DIRBLK), &g.nodeof(0), seq_lds };
    unsigned kinds[5] = {0, 0, 0, 0, 0};
    const bool ph_detail = POA_PHC(J) && J.phase_detail;
#if POA_REPEAT == 1
    for (int rep_ = 0; rep_ < 2; ++rep_) {
#endif
    // band start per ANCHOR (anchors are coordinates in the first sequence: 0 .. L0-1): one division per anchor value in a table (the direction block is free
    // until the forward pass stages its first row) instead of three per rank
    const bool lt_lds = POA_LT && (unsigned)st.L0 * 2u <= (unsigned)(TBR * BW);
    const l16 lt = POA_LDS(l16, LLT<BW>::DIRBLK);
    if (lt_lds) { for (int a = lane; a < st.L0; a += 64) lt[a] = (uint16_t)band_lo(a, bm, BW); lds_sync(); }
    auto BL = [&](int anchor) __attribute__((always_inline)) -> int { return lt_lds ? (int)lt[anchor] : band_lo(anchor, bm, BW); };
    for (int rb = 0; rb < V; rb += 128) {
        uint32_t ppv[2], arv[2]; int cmv[2], ofv[2], frv[2], lp0[2], lp1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int r = rb + u * 64 + lane; const bool ok = r < V; ppv[u] = ok ? g.pp(r) : 0xFFFFFFFFu; arv[u] = ok ? g.ar(r) : 0u; cmv[u] = ok ? (int)g.cm(r) : 0; ofv[u] = ok ? (int)g.of(r) : 1; frv[u] = ok ? (int)g.far(r) : 0; }
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int a = ppv[u] & 0xffff, b = ppv[u] >> 16; lp0[u] = a != NONE16 ? (int)g.anchor(a) : 0; lp1[u] = b != NONE16 ? (int)g.anchor(b) : 0; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = rb + u * 64 + lane; const bool ok = r < V;          // (no divergent exits: the run lengths below are a wave ballot)
            const int p0 = ppv[u] & 0xffff, p1 = ppv[u] >> 16;
            const int l0 = BL((int)(arv[u] & 0xffff)); int fl = 0, d0 = 0, d1 = 0, dl0 = 0, dl1 = 0;
            if (ok) {
                if (p0 == NONE16) fl |= 1;
                else {
                    d0 = r - p0; dl0 = l0 - BL(lp0[u]);
                    if (p1 != NONE16) { d1 = r - p1; dl1 = l0 - BL(lp1[u]); if (cmv[u] & 0x80) fl |= 2; }
                    if (d0 > 255 || d1 > 255 || dl0 < 0 || dl0 > 255 || dl1 < 0 || dl1 > 255) { fl |= 2; d0 = d1 = dl0 = dl1 = 0; }
                }
                if (!ofv[u]) fl |= 4;
                if (frv[u]) fl |= 8;
                if (!(fl & 3)) {
                    if (d0 == 1 && d1 == 0 && dl0 <= 1) fl |= 16 | (dl0 << 5);
                    else if (d0 <= HR && d1 <= HR && dl0 <= DLO_MAX && dl1 <= DLO_MAX) fl |= 64;
                }
            }
            // TIGHT runs of the forward pass: consecutive chain rows whose band moves by one column per row, that hold no end cell, need no HBM copy and lie
            // in one block of direction rows.  Every row of a run carries the number of rows left in it (flag 128, count in the dlo0 byte, which chain rows
            // do not use): the forward pass does such a run in a counted loop without decoding flags row by row.
            const bool endz = (mode == NGSID_POA_SEMI || (fl & 4)) && (unsigned)(L - l0) < (unsigned)BW;
            const bool tight = ok && (fl & (16 | 32 | 8)) == (16 | 32) && (mode == NGSID_POA_LOCAL || !endz);
            const unsigned long long tm = __ballot(tight);
            if (tight) {
                const unsigned long long x = ~(tm >> lane);
                int cr = x ? __builtin_ctzll(x) : 64;
                cr = min(cr, TBR - (lane & (TBR - 1)));
                dl0 = cr; fl |= 128;
            }
            if (ok) g.ri(r) = (unsigned long long)(unsigned)l0 | ((unsigned long long)(unsigned)d0 << 16) | ((unsigned long long)(unsigned)d1 << 24)
                               | ((unsigned long long)(unsigned)dl0 << 32) | ((unsigned long long)(unsigned)dl1 << 40)
                               | ((unsigned long long)(unsigned)(cmv[u] & 0x7f) << 48) | ((unsigned long long)(unsigned)fl << 56);
            if (ph_detail) {            // dev instrumentation: row kinds of the forward pass (counted per alignment, one atomic each)
                const unsigned long long tmk = tm;
                kinds[0] += __popcll(tmk); kinds[1] += __popcll(__ballot(tight && (lane == 0 || (lane & (TBR - 1)) == 0 || !((tmk >> (lane - 1)) & 1))));
                kinds[2] += __popcll(__ballot(ok && !tight && (fl & 16))); kinds[3] += __popcll(__ballot(ok && !(fl & 16) && (fl & 64))); kinds[4] += __popcll(__ballot(ok && !(fl & (16 | 64))));
            }
        }
    }
    for (int i = lane; i < HR * (RPADL + RPADR); i += 64) {       // guard cells of the ring rows
        const int row = i / (RPADL + RPADR), k = i % (RPADL + RPADR);
        w.hring()[row * (BW + RPADL + RPADR) + (k < RPADL ? k : BW + k)] = 0;          // "minus infinity" of the biased cell values
    }
    for (int i = lane; i < L; i += 64) { if (!seq_lds) g.alnode(i) = NONE16; w.sq()[i] = S.s[i]; }
    for (int i = lane; i < BW; i += 64) w.sq()[L + i] = 0xFF;            // pad: columns past the end never match
    if (lane == 0) w.sq()[-1] = 0xFF;
#if POA_REPEAT == 1
    mem_sync();
    }
#endif
    if (ph_detail && lane == 0) for (int k = 0; k < 5; ++k) atomicAdd(&PHS(J)[16 + k], (unsigned long long)kinds[k]);
    mem_sync();
    PH(J, 0, tph);
    // ---------- forward DP, one row per graph node in topological order
    const bool local = mode == NGSID_POA_LOCAL;
    int bestv, bestpk, nslow;
#if POA_REPEAT == 2
    for (int rep_ = 0; rep_ < 2; ++rep_)
#endif
#if POA_FWD_CALL
    { FwdOut fo;
      if (local) fo = poa_forward_call<CPL, NGSID_POA_LOCAL>((unsigned long long)g.base, g.s32, g.s64, g.s16, g.s8, g.se16, g.se32, g.sl, (unsigned long long)Hg, (unsigned long long)Dg, (unsigned long long)Dfull, L, V, st.nov, gp, J.m, J.n, lane);
      else if (mode == NGSID_POA_SEMI) fo = poa_forward_call<CPL, NGSID_POA_SEMI>((unsigned long long)g.base, g.s32, g.s64, g.s16, g.s8, g.se16, g.se32, g.sl, (unsigned long long)Hg, (unsigned long long)Dg, (unsigned long long)Dfull, L, V, st.nov, gp, J.m, J.n, lane);
      else fo = poa_forward_call<CPL, NGSID_POA_GLOBAL>((unsigned long long)g.base, g.s32, g.s64, g.s16, g.s8, g.se16, g.se32, g.sl, (unsigned long long)Hg, (unsigned long long)Dg, (unsigned long long)Dfull, L, V, st.nov, gp, J.m, J.n, lane);
      bestv = fo.bestv; bestpk = fo.bestpk; nslow = fo.nslow; }
#else
    if (local) poa_forward<CPL, NGSID_POA_LOCAL>(g, w, Hg, Dg, Dfull, S, V, st.nov, gp, J.m, J.n, lane, bestv, bestpk, nslow);
    else if (mode == NGSID_POA_SEMI) poa_forward<CPL, NGSID_POA_SEMI>(g, w, Hg, Dg, Dfull, S, V, st.nov, gp, J.m, J.n, lane, bestv, bestpk, nslow);
    else poa_forward<CPL, NGSID_POA_GLOBAL>(g, w, Hg, Dg, Dfull, S, V, st.nov, gp, J.m, J.n, lane, bestv, bestpk, nslow);
#endif
    if (POA_PHC(J) && lane == 0) { atomicAdd(&PHS(J)[5], (unsigned long long)V); atomicAdd(&PHS(J)[6], (unsigned long long)nslow); }
    mem_sync();                                       // direction rows must have landed before the traceback pulls them back
    PH(J, 1, tph);
    // ---------- best end cell: max value, ties -> lowest rank, then lowest column
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int ov = __shfl_xor(bestv, d), opk = __shfl_xor(bestpk, d);
        if (ov > bestv || (ov == bestv && opk < bestpk)) { bestv = ov; bestpk = opk; }
    }
    const int bestr = bestpk == 0x7fffffff ? -1 : (bestpk >> 8), bestc = bestpk & 0xff;
    if (POA_PHC(J) && lane == 0) { atomicAdd(&PHS(J)[8], (unsigned long long)(unsigned)bestv); atomicAdd(&PHS(J)[9], (unsigned long long)(unsigned)bestpk); }
    bool aligned_any = true;
    if (bestr < 0 || (mode == NGSID_POA_LOCAL && bestv <= 0)) {
        if (mode != NGSID_POA_LOCAL) return 0;
        aligned_any = false;                                   // nothing aligned: the whole read becomes a new branch
    }
    // ---------- traceback in rank space: predecessors and band starts come from the row info in LDS, direction rows are pulled TBR at
    //            a time from HBM into LDS.  alnode[] receives RANKS here.  The walk is serial, but most of it is runs of plain
    //            diagonal moves through chain rows (predecessor = previous rank): lane k speculatively inspects the cell k such moves
    //            ahead, the wave takes the whole leading run at once, and the first other move is decoded from that lane's data.
#if POA_REPEAT == 3
    for (int rep_ = 0; rep_ < 2; ++rep_) {
#endif
    if (seq_lds) { for (int i = lane; i < L; i += 64) alnode.l[i] = NONE16; lds_sync(); }      // (the ring is free now)
    if (aligned_any) {
        // r, j: wave-uniform (scalar registers: the whole walk is scalar control flow, only the speculative look-ahead is per lane)
        int r = __builtin_amdgcn_readfirstlane(bestr);
        int j = (int)(__builtin_amdgcn_readfirstlane((unsigned)g.ri(r)) & 0xffff) + __builtin_amdgcn_readfirstlane(bestc);
        int blk_lo = (V - 1) & ~(TBR - 1);                 // the forward pass left the last block of direction rows in LDS, packed
#if POA_REPEAT == 3
        if (rep_) blk_lo = 0x7fffff00;                     // timing build: the second walk reloads its first block from HBM (the forward pass flushed it)
#endif
        // lane k keeps the row info of rank blk_lo + k in registers (k < TBR): one LDS read per iteration (the direction nibble) instead of two
        unsigned long long myri = lane < TBR ? w.rblk()[lane] : 0ull;
        int n_reload = 0, n_iter = 0; unsigned long long c_reload = 0;
        // the block below the current one is prefetched into registers while the current one is walked (a reload is an L2 / HBM round trip
        // of several thousand cycles, the walk of a block takes longer than that).  Blocks stay PACKED in LDS (4 bits per cell: type | slot
        // code << 2): a cell costs one byte read and a shift, a reload one 16-byte LDS store per piece and no unpacking.
        constexpr int NPF = TBR * BW / 32 / 64;                // packed 16-byte pieces per lane and block
        const l8 pkblk = w.dirblk();
        ngsid_v4u pf[NPF]; unsigned long long pri = 0ull; int pf_blk = -1;
        auto prefetch = [&](int b) {
            if (b < 0) { pf_blk = -1; return; }
            const uint8_t* src = Dg + (size_t)b * (BW / 2);
#pragma unroll
            for (int x = 0; x < NPF; ++x) pf[x] = ngsid_load16_l2(src + (lane + 64 * x) * 16);       // L2-served: rows are rewritten per sequence
            pri = lane < TBR ? g.ri(b + lane) : 0ull;
            pf_blk = b;
        };
        int edge = 0;
#if POA_REPEAT == 3
        if (rep_) pf_blk = -1; else
#endif
        prefetch(blk_lo - TBR);
        for (int guard = 0;; ++guard) {
            ++n_iter;
            if (guard > 2 * (V + L) + 64) { if (lane == 0 && J.slot_overflow) atomicExch(J.slot_overflow + 1, 1u); break; }   // cannot happen: every iteration consumes a move (reported by the host as an internal error)
            if (r < blk_lo) {
                lds_sync();
                blk_lo = r & ~(TBR - 1);                    // aligned blocks of TBR rows, same layout as the forward pass staged them
                ++n_reload;
                const unsigned long long trl0 = POA_PHC(J) ? __builtin_readcyclecounter() : 0;
                if (blk_lo != pf_blk) prefetch(blk_lo);       // the path jumped further than one block (far predecessor): fetch it now
#pragma unroll
                for (int x = 0; x < NPF; ++x) *(LDSP ngsid_v4u*)(pkblk + (size_t)(lane + 64 * x) * 16) = pf[x];
                myri = pri;
                prefetch(blk_lo - TBR);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (POA_PHC(J)) c_reload += __builtin_readcyclecounter() - trl0;
            }
            // lane k looks at row blk_lo + k: the cell the path reaches there if it only takes diagonal moves through chain rows from (r, j)
            const int top = r - blk_lo;                        // 0 .. TBR-1
            const int jk = j - (top - lane);
            // branch-free per-lane part: out-of-block / out-of-band lanes read cell 0 and are masked afterwards
            const int lok = (int)(myri & 0xffff);
            const int ck = jk - lok;
            const bool loaded = (lane <= top) & ((unsigned)ck < (unsigned)BW);
            const int pb = pkblk[loaded ? lane * (BW / 2) + (ck >> 1) : 0];
            const int nib = (pb >> ((ck & 1) * 4)) & 15;
            // a plain diagonal move to the previous rank: nibble 0 = diagonal through the first in-edge, whose tail is one rank up (dist0 == 1; true
            // for every chain row and for the near rows whose first predecessor is the previous rank).  Rows with more than two in-edges (flag 2)
            // keep their real bytes in HBM: never part of a run, decoded below.
            const bool good = loaded & (nib == 0) & (((unsigned)(myri >> 16) & 0xffu) == 1u) & !((unsigned)(myri >> 56) & 2u) & (jk >= 1);
            const unsigned long long gm = __ballot(good);
            const unsigned long long x = gm << (63 - top);     // lane `top` at bit 63: leading ones = the run
            const int run = (~x) ? __builtin_clzll(~x) : 64;   // <= top + 1 because lanes above `top` never set their bit
            {   // band-edge check (oracle poa_align): a visited cell (the run and the cell it ends on) on a clipped edge of its row's band
                const bool clipped = ((ck == 0) & (lok > 0)) | ((ck == BW - 1) & (lok + BW - 1 < L));
                if (__ballot(loaded & (lane >= top - run) & clipped)) edge = 1;
            }
            if ((lane <= top) & (lane > top - run)) alnode.set(jk - 1, blk_lo + lane);      // `run` diagonal moves, each to the previous rank
            r -= run; j -= run;
            const int nk = top - run;                          // lane holding the next cell of the path
            if (nk < 0) continue;                              // it is in the block below: go round (loads it)
            const unsigned rlo = __builtin_amdgcn_readlane((unsigned)myri, nk), rhi = __builtin_amdgcn_readlane((unsigned)(myri >> 32), nk);
            int type, slot;
            if ((rhi >> 24) & 2) {                             // more than two in-edges: the real direction byte
                const int d = __builtin_amdgcn_readfirstlane((int)Dfull[(size_t)r * BW + (j - (int)(rlo & 0xffff))]);
                type = d & 3; slot = d >> 2;
            } else {
                const int nb = __builtin_amdgcn_readlane(nib, nk);
                type = nb & 3; slot = (nb & 8) ? SRC_SLOT : ((nb >> 2) & 1);
            }
            if (type == 3) break;
            if (type == 2) { --j; continue; }
            if (type == 0) { if (lane == 0) alnode.set(j - 1, r); --j; }
            if (slot == SRC_SLOT) break;
            int pr;
            if (!((rhi >> 24) & 2) && slot <= 1) pr = r - (int)(slot == 0 ? ((rlo >> 16) & 0xff) : (rlo >> 24));
            else if (slot <= 1) pr = __builtin_amdgcn_readfirstlane(slot == 0 ? (int)g.p0(r) : (int)g.p1(r));
            else { int e = -1; for (int t = 2; t <= slot; ++t) e = ov_next(g, st.nov, r, e + 1, lane); pr = __builtin_amdgcn_readfirstlane((int)g.ov_tail(e)); }
            r = pr;
        }
        if (POA_PHC(J) && lane == 0) { atomicAdd(&PHS(J)[7], (unsigned long long)n_iter); atomicAdd(&PHS(J)[13], (unsigned long long)n_reload); atomicAdd(&PHS(J)[14], c_reload); }
        edge_out |= edge;
    }
#if POA_REPEAT == 3
    lds_sync();
    }
#endif
    if (seq_lds) lds_sync(); else mem_sync();         // alnode[] is read by other lanes next
    PH(J, 2, tph);
    // ---------- A: the existing node per position: the aligned node if the letter matches, else a sibling (same column) with that letter whose rank lies
    //             strictly between the previous aligned position's node and this one (oracle g_add_alignment / rg_add_alignment: keeps the order
    //             topological without a re-sort).  alnode[] keeps the aligned RANK, nodeof[] receives the existing rank or NONE16 (= a new node).
    // Per 64-position chunk phase A leaves a summary in LDS (the DP ring is free now): the mask of positions that need a NEW node, the aligned
    // rank of the chunk's last aligned position and the rank CHOSEN for its first aligned position (reused sibling or the aligned node).
    const unsigned ch_base = seq_lds ? LLT<BW>::RBLK : 0u;                  // (the ring and the direction block hold alnode[] / nodeof[] then)
    const lu64 ch_new = POA_LDS(lu64, ch_base);                             // [nch]
    const l16 ch_last = POA_LDS(l16, ch_base + 8u * (unsigned)((L + 63) / 64));      // [nch] aligned rank of the last aligned position, NONE16 = none
    const l16 ch_first = ch_last + (L + 63) / 64;                          // [nch] rank chosen for the first aligned position, NONE16 = none
    const int nch = (L + 63) / 64;
    int nnew = 0;
    {
        int carry = -1;                                   // aligned rank of the nearest aligned position before the chunk
        for (int ib = 0; ib < L; ib += 128) {             // two chunks per iteration: both chunks' loads are in flight together
            int arv[2], cdv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int i = ib + u * 64 + lane; arv[u] = (i < L) ? alnode.get(i) : NONE16; }
#pragma unroll
            for (int u = 0; u < 2; ++u) cdv[u] = arv[u] != NONE16 ? (int)(g.cm(arv[u]) & 0x7f) : -1;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = ib + u * 64 + lane; bool isnew = false;
                const int ar = arv[u];
                const unsigned long long ma = __ballot(ar != NONE16);
                const unsigned long long lt = ma & ((1ull << lane) - 1);
                const int psrc = lt ? 63 - __clzll(lt) : 0;
                const int pv = __shfl(ar, psrc);
                const int prev_rank = lt ? pv : carry;
                int chosen = NONE16;
                if (i < L) {
                    const int ch = w.sq()[i]; int found = NONE16;
                    if (ar != NONE16) {
                        if (cdv[u] == ch) found = ar;
                        else for (int x = g.ring(ar); x != ar; x = g.ring(x)) if ((int)(g.cm(x) & 0x7f) == ch && x > prev_rank && x < ar) { found = x; break; }
                        chosen = found != NONE16 ? found : ar;
                    }
                    nodeof.set(i, found); isnew = found == NONE16;
                }
                const unsigned long long mn = __ballot(isnew);
                nnew += __popcll(mn);
                if (ib + u * 64 < L) {
                    const int fc = __shfl(chosen, ma ? __ffsll((long long)ma) - 1 : 0), lv = __shfl(ar, ma ? 63 - __clzll(ma) : 0);
                    if (lane == 0) { const int c = (ib >> 6) + u; ch_new[c] = mn; ch_first[c] = (uint16_t)(ma ? fc : NONE16); ch_last[c] = (uint16_t)(ma ? lv : NONE16); }
                }
                if (ma) { const int hl = 63 - __clzll(ma); carry = __shfl(ar, hl); }
            }
        }
    }
    if (ph_detail) { unsigned long long sm_ = 0; for (int i = lane; i < L; i += 64) sm_ += (unsigned long long)(alnode.get(i) + 1) * (unsigned)(i + 1); for (int d = 32; d >= 1; d >>= 1) sm_ += __shfl_xor(sm_, d); if (lane == 0) { atomicAdd(&PHS(J)[10], (unsigned long long)nnew); atomicAdd(&PHS(J)[11], sm_); } }
    if (seq_lds) lds_sync(); else mem_sync();
    unsigned long long tpu = tph; PH(J, 21, tpu);
    if (V + nnew > st.capV || st.E + L > st.capE) return 2;      // oracle g_add_alignment capacity rule
    // ---------- new nodes of the chunks that have any (ids do not exist any more: a node IS its rank).  The k-th new node of the sequence goes immediately
    //             before old rank ins = the rank chosen for the first aligned position at or after its own (V = the end; non-decreasing in k) and lands on
    //             rank ins + k.  One routine, two uses: pass 0 only needs (ins, k) to build shift[]; pass 1, after the old records have moved, writes the records.
    // shift[r] (r <= V) = new nodes inserted at or before old rank r: an old node moves from r to r + shift[r].  u8 in the LDS behind alnode[] when the counts
    // fit a byte and the ring has the room, else u16 in the HBM scratch.
    const bool sh_lds = seq_lds && nnew <= 255 && (unsigned)(2 * ((L + 7) & ~7) + V + 2) <= (unsigned)(HR * (BW + RPADL + RPADR) * 4);
    const l8 sh_l = POA_LDS(l8, LLT<BW>::HRING + 2u * (unsigned)((L + 7) & ~7));
    // compact list of the new nodes (position, aligned rank, nearest aligned rank at or before, reference rank) behind nodeof[] in the direction block: the
    // records of ALL new nodes are then written in one round (one dependent round trip for the anchors / ring links instead of one per 64-position chunk)
    const int nn_cap = seq_lds ? (int)((unsigned)(TBR * BW) - 2u * (unsigned)((L + 7) & ~7)) / 8 : 0;
    const bool nn_lds = nnew > 0 && nnew <= nn_cap;
    const l16 nn_l = POA_LDS(l16, LLT<BW>::DIRBLK + 2u * (unsigned)((L + 7) & ~7));      // [4][nn_cap]
    auto SH = [&](int x) __attribute__((always_inline)) -> int { return sh_lds ? (int)sh_l[x] : (int)g.shiftg(x); };
    auto RM = [&](int x) __attribute__((always_inline)) -> int { return x + SH(x); };
    auto new_nodes = [&](const int pass) __attribute__((always_inline)) {
        int base = 0, lastal = NONE16;
        for (int c = 0; c < nch; ++c) {
            const unsigned long long mn = ch_new[c];
            if (mn) {
                const int i = c * 64 + lane; const bool isnew = (mn >> lane) & 1ull;
                const int a = (i < L) ? alnode.get(i) : NONE16; const int nf = (i < L && !isnew) ? nodeof.get(i) : NONE16;
                const int chosen = a != NONE16 ? (nf != NONE16 ? nf : a) : NONE16;
                const unsigned long long ma = __ballot(a != NONE16);
                const unsigned long long le = ma & (~0ull >> (63 - lane));
                const int lv = __shfl(a, le ? 63 - __clzll(le) : 0);
                const int la = le ? lv : lastal;                   // nearest aligned position at or before i
                const unsigned long long ge = ma & (~0ull << lane);
                const int rv = __shfl(chosen, ge ? __ffsll((long long)ge) - 1 : 0);
                int nxt = NONE16;                                  // rank chosen for the first aligned position after the chunk
                for (int c2 = c + 1; c2 < nch; ++c2) { const int f = ch_first[c2]; if (f != NONE16) { nxt = f; break; } }
                const int rf = ge ? rv : nxt;
                const int k = base + __popcll(mn & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                const int ins = rf != NONE16 ? rf : V;
                if (pass == 0) {
                    // the LAST new node of a run of equal insertion points writes the count; a later chunk with the same point overwrites it with a larger one
                    const unsigned long long later = mn & ((lane == 63) ? 0ull : (~0ull << (lane + 1)));
                    const int ins_next = __shfl(ins, later ? __ffsll((long long)later) - 1 : 0);
                    if (isnew && (!later || ins_next != ins)) { if (sh_lds) sh_l[ins] = (uint8_t)(k + 1); else g.shiftg(ins) = (uint16_t)(k + 1); }
                    if (isnew && nn_lds) { nn_l[k] = (uint16_t)i; nn_l[nn_cap + k] = (uint16_t)a; nn_l[2 * nn_cap + k] = (uint16_t)la; nn_l[3 * nn_cap + k] = (uint16_t)rf; }
                    if (!sh_lds) mem_sync();                        // (HBM fall-back: keep the stores of successive chunks to one address in order)
                } else if (isnew) {
                    const int y = ins + k;
                    const int anc = la != NONE16 ? (int)g.anchor(RM(la)) : (rf != NONE16 ? (int)g.anchor(RM(rf)) : (S.a1 < S.a0 ? 0 : S.a0));
                    int rg = y;
                    if (a != NONE16) { const int v = RM(a); rg = g.ring(v); g.ring(v) = (uint16_t)y; }      // joins the ring right behind the node it is aligned to
                    g.cm(y) = w.sq()[i]; g.ar(y) = (uint32_t)anc | ((uint32_t)rg << 16); g.pp(y) = 0xFFFFFFFFu; g.ww(y) = 0ull; g.cov(y) = 0u; g.of(y) = 0;
                    nodeof.set(i, y);
                }
                base += __popcll(mn);
            }
            const int lc = ch_last[c];
            if (lc != NONE16) lastal = lc;
        }
    };
    if (nnew) {
        // ---------- S: shift[] = running maximum of the counts the runs wrote; far[] of the whole new rank range is cleared (the move recomputes it)
        for (int r = lane; r <= V; r += 64) { if (sh_lds) sh_l[r] = 0; else g.shiftg(r) = 0; }
        for (int r = lane; r < V + nnew; r += 64) g.far(r) = 0;
        if (sh_lds) lds_sync(); else mem_sync();
        new_nodes(0);
        if (sh_lds) lds_sync(); else mem_sync();
        {
            int carry = 0;
            for (int rb = 0; rb <= V; rb += 64) {
                const int r = rb + lane; const int v = r <= V ? SH(r) : 0;
                const int m = max(wave_incl_max_scan(v), carry);
                if (r <= V) { if (sh_lds) sh_l[r] = (uint8_t)m; else g.shiftg(r) = (uint16_t)m; }
                carry = __builtin_amdgcn_readlane(m, 63);
            }
        }
        mem_sync();                                       // (also: far[] cleared before the move sets it)
        // ---------- D: the old records move up by shift[], highest ranks first (in place: a destination never lies below its source, distinct ranks have
        //             distinct destinations, and both chunks of an iteration are loaded before either is stored); rank-valued fields are remapped; a
        //             predecessor that ends up more than HR ranks before its successor gets its far flag back
        for (int rb = (V - 1) & ~127; rb >= 0; rb -= 128) {
            uint32_t ppv[2], arv[2], cvv[2]; unsigned long long wwv[2]; int cmv[2], ofv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int r = rb + u * 64 + lane; if (r < V) { ppv[u] = g.pp(r); arv[u] = g.ar(r); cvv[u] = g.cov(r); wwv[u] = g.ww(r); cmv[u] = g.cm(r); ofv[u] = g.of(r); } else { ppv[u] = 0xFFFFFFFFu; arv[u] = 0; cvv[u] = 0; wwv[u] = 0; cmv[u] = 0; ofv[u] = 0; } }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = rb + u * 64 + lane;
                if (r < V) {
                    const int nr = RM(r); int a = ppv[u] & 0xffff, b = ppv[u] >> 16; const int rg = RM((int)(arv[u] >> 16));
                    if (a != NONE16) { a = RM(a); if (nr - a > HR) g.far(a) = 1; }
                    if (b != NONE16) { b = RM(b); if (nr - b > HR) g.far(b) = 1; }
                    g.pp(nr) = (uint32_t)a | ((uint32_t)b << 16); g.ar(nr) = (arv[u] & 0xffffu) | ((uint32_t)rg << 16); g.cov(nr) = cvv[u]; g.ww(nr) = wwv[u]; g.cm(nr) = (uint8_t)cmv[u]; g.of(nr) = (uint8_t)ofv[u];
                }
            }
        }
        for (int x = lane; x < st.nov; x += 64) { const int h = RM((int)g.ov_head(x)), t = RM((int)g.ov_tail(x)); g.ov_head(x) = (uint16_t)h; g.ov_tail(x) = (uint16_t)t; if (h - t > HR) g.far(t) = 1; }
        mem_sync();
        PH(J, 22, tpu);
        // ---------- N: records of the new nodes
        if (nn_lds) {
            for (int k0 = 0; k0 < nnew; k0 += 64) {
                const int k = k0 + lane;
                if (k < nnew) {
                    const int i = nn_l[k], a = nn_l[nn_cap + k], la = nn_l[2 * nn_cap + k], rf = nn_l[3 * nn_cap + k];
                    const int y = (rf != NONE16 ? rf : V) + k;
                    const int anc = la != NONE16 ? (int)g.anchor(RM(la)) : (rf != NONE16 ? (int)g.anchor(RM(rf)) : (S.a1 < S.a0 ? 0 : S.a0));
                    int rg = y;
                    if (a != NONE16) { const int v = RM(a); rg = g.ring(v); g.ring(v) = (uint16_t)y; }      // joins the ring right behind the node it is aligned to
                    g.cm(y) = w.sq()[i]; g.ar(y) = (uint32_t)anc | ((uint32_t)rg << 16); g.pp(y) = 0xFFFFFFFFu; g.ww(y) = 0ull; g.cov(y) = 0u; g.of(y) = 0;
                    nodeof.set(i, y);
                }
            }
        } else new_nodes(1);
        // existing nodes the sequence goes through: their final ranks
        for (int i = lane; i < L; i += 64) if (!((ch_new[i >> 6] >> (i & 63)) & 1ull)) nodeof.set(i, RM(nodeof.get(i)));
        mem_sync();
        PH(J, 23, tpu);
    }
    // ---------- E: coverage and edges along the sequence (final ranks).  Every node of the path is touched by ONE position (its in-edge record, its coverage);
    //             the out-flag and the far flag of a node are written by the lane of the NEXT position (separate byte arrays).  A new edge takes the first
    //             free inline slot of its head, else an overflow entry (sequence order)
    {
        int ebase = st.E, nov = st.nov;
        for (int ib = 0; ib < L; ib += 128) {             // two chunks per iteration (independent)
            int av[2], bv[2]; uint32_t ppb[2], covv[2]; unsigned long long wwb[2]; int cmb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int i = ib + u * 64 + lane; bv[u] = i < L ? nodeof.get(i) : NONE16; av[u] = (i < L && i > 0) ? nodeof.get(i - 1) : NONE16; }
#pragma unroll
            for (int u = 0; u < 2; ++u) { const bool h = bv[u] != NONE16; ppb[u] = h ? g.pp(bv[u]) : 0xFFFFFFFFu; wwb[u] = h ? g.ww(bv[u]) : 0ull; covv[u] = h ? g.cov(bv[u]) : 0u; cmb[u] = h ? (int)g.cm(bv[u]) : 0; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int i = ib + u * 64 + lane; bool newedge = false, needov = false; const int a = av[u], b = bv[u]; int wgt = 0;
                if (i < L) {
                    g.cov(b) = covv[u] + S.cw;
                    if (i > 0) {
                        wgt = wtof(S, i - 1) + wtof(S, i);
                        const int t0 = ppb[u] & 0xffff, t1 = ppb[u] >> 16;
                        if (t0 == a) g.w0(b) = (int)(unsigned)wwb[u] + wgt;
                        else if (t1 == a) g.w1(b) = (int)(unsigned)(wwb[u] >> 32) + wgt;
                        else {
                            int e = -1;
                            if (cmb[u] & 0x80) for (int x = 0; x < nov; ++x) if ((int)g.ov_head(x) == b && (int)g.ov_tail(x) == a) { e = x; break; }
                            if (e >= 0) g.ov_w(e) += wgt;
                            else {
                                newedge = true;
                                if (t0 == NONE16) { g.p0(b) = (uint16_t)a; g.w0(b) = wgt; }
                                else if (t1 == NONE16) { g.p1(b) = (uint16_t)a; g.w1(b) = wgt; }
                                else needov = true;
                                g.of(a) = 1; if (b - a > HR) g.far(a) = 1;
                            }
                        }
                    }
                }
                const unsigned long long mo = __ballot(needov);
                if (needov) {
                    const int e = nov + __popcll(mo & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
                    g.ov_head(e) = (uint16_t)b; g.ov_tail(e) = (uint16_t)a; g.ov_w(e) = wgt; g.cm(b) = (uint8_t)(cmb[u] | 0x80);
                }
                if (mo) { nov += __popcll(mo); mem_sync(); }       // (a later chunk may look the list up)
                ebase += __popcll(__ballot(newedge));
            }
        }
        st.E = ebase; st.nov = nov;
    }
    st.V = V + nnew; st.cw_sum += S.cw;
    mem_sync();
    PH(J, 3, tph);
    return 1;
}

// LDS bytes of one tile (must match the carve in k_poa_tile)
size_t poa_lds_bytes(int Vc, int Ec, int Lm, int BW)
{
    (void)Ec;
    const size_t aln = (size_t)HR * (BW + RPADL + RPADR) * 4 + (size_t)TBR * BW + (size_t)TBR * 8 + poa_al16((size_t)Lm + BW + 32);
    const size_t cons = poa_al16((size_t)2 * Vc) + poa_al16(((size_t)Vc + 31) / 32 * 4);
    return aln > cons ? aln : cons;
}
// HBM scratch bytes of one workgroup for the graph arrays
static size_t poa_graph_bytes(int Vc, int Ec, int Lm)
{
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t n = (size_t)Vc + 1;
    return 3 * al(4 * n) + 3 * al(8 * n) + 2 * al(2 * n) + 3 * al(n) + 2 * al(2 * (size_t)Ec) + al(4 * (size_t)Ec) + 2 * al(2 * (size_t)Lm);
}

template <int CPL>
__device__ __forceinline__ void poa_tile_body(const PoaJobSet& J, uint8_t* gscratch, size_t gbytes, uint32_t* __restrict__ work_ctr)
{
    constexpr int BW = 64 * CPL;
    const int lane = threadIdx.x;
    const int Vc = J.Vcap, Ec = J.Ecap, Lm = J.Lmax;
    LLT<BW> w; GG g;
    {
        auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
        w.sinkbits = POA_LDS(LDSP unsigned int*, (unsigned)al((size_t)2 * Vc));
        g.base = gscratch + (size_t)blockIdx.x * gbytes;
        g.s32 = (uint32_t)al(4 * ((size_t)Vc + 1)); g.s64 = (uint32_t)al(8 * ((size_t)Vc + 1)); g.s16 = (uint32_t)al(2 * ((size_t)Vc + 1)); g.s8 = (uint32_t)al((size_t)Vc + 1);
        g.se16 = (uint32_t)al(2 * (size_t)Ec); g.se32 = (uint32_t)al(4 * (size_t)Ec); g.sl = (uint32_t)al(2 * (size_t)Lm);
    }
    int32_t* Hg = J.Hglob + (size_t)blockIdx.x * Vc * BW;
    uint8_t* Dg = J.dirglob + (size_t)blockIdx.x * Vc * BW * 3 / 2;      // packed direction blocks (4 bits per cell) ...
    uint8_t* Dfull = Dg + (size_t)Vc * BW / 2;                            // ... and the byte rows of the ranks with more than two in-edges (sparse)

#if POA_COLD
    // Round 5: the fields of the job description that only the per-tile set-up and the emission read (15 pointers: sequence / tile lists, output arrays, flags) are
    // read from the kernel-argument segment WHERE they are used, through a pointer the compiler cannot see through - kept in SGPRs across the alignment phases they
    // were part of the 172 spilled SGPRs of the 64-column instance (v_writelane / v_readlane pairs = VALU instructions in the row paths).  J is the first kernel argument.
    const PoaJobSet* Jc_ = (const PoaJobSet*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(Jc_));
    const PoaJobSet& JC = *Jc_;
#else
    const PoaJobSet& JC = J;
#endif
    const uint32_t nrun_ = JC.nrun_dev ? (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_nontemporal_load(JC.nrun_dev)) : JC.nrun;      // (written by an earlier kernel of the stream)
    for (;;) {
        // persistent workgroups pull tiles from a queue (tiles differ a lot in cost: depth, graph growth, splits)
        uint32_t jq = 0; if (lane == 0) jq = atomicAdd(work_ctr, 1u);
        const uint32_t jqi = (uint32_t)__builtin_amdgcn_readfirstlane((int)jq);
        if (jqi >= nrun_) break;
        const uint32_t job = JC.job_list ? JC.job_list[jqi] : jqi;          // a redo launch (wider band) runs a list of tiles
        int edge = 0;
        const uint32_t s0 = JC.job_off[job], s1 = JC.job_off[job + 1];
        const int bbi = JC.job_bb ? JC.job_bb[job] : -1;
        TS st; st.V = 0; st.E = 0; st.L0 = 0; st.members = 0; st.nout = 0; st.cw_sum = 0; st.nov = 0;
        uint32_t ndrop = 0; unsigned long long nrows = 0;
        {   // per-job capacity = oracle run_tile: cap_for(L0) but at least the longest member + 1; edges 1.5x
            int maxlen = bbi >= 0 ? JC.bbs[bbi].len : 0, first = bbi >= 0 ? JC.bbs[bbi].len : 0;
            for (uint32_t si = s0; si < s1; ++si) { const int l = JC.seqs[JC.seq_idx ? JC.seq_idx[si] : si].len; if (l > maxlen) maxlen = l; if (first == 0 && bbi < 0 && si == s0) first = l; }
            long long c = (long long)(first > 0 ? first : 1) * (JC.node_cap > 0 ? JC.node_cap : 28) / 16; if (c < (first > 0 ? first : 1) + 64) c = (first > 0 ? first : 1) + 64;
            if (c < maxlen + 1) c = maxlen + 1;
            st.capV = (int)(c < Vc ? c : Vc); st.capE = 3 * st.capV / 2 < Ec ? 3 * st.capV / 2 : Ec;
        }
        for (uint32_t si = s0; si < s1; ++si) {
            const PSeq S = JC.seqs[JC.seq_idx ? JC.seq_idx[si] : si];
            if (S.len <= 0) continue;
            if (S.len > Lm) { ++ndrop; continue; }
            // at most two attempts: when the graph is full (code 2) the tile is emitted and the sequence starts / joins a fresh one
            for (int attempt = 0; attempt < 2; ++attempt) {
                if (st.V == 0) {
                    if (bbi >= 0) { const PSeq B = JC.bbs[bbi]; tile_add_first(g, B, st, lane); }
                    else { if (S.len > st.capV) ++ndrop; else { tile_add_first(g, S, st, lane); st.members = 1; } break; }
                }
                nrows += (unsigned)st.V;
                const int rcode = tile_align_add<CPL>(g, w, Hg, Dg, Dfull, J, S, st, lane, edge);
                if (rcode == 1) { st.members += 1; break; }
                if (rcode == 0 || attempt == 1) { ++ndrop; break; }
                tile_emit(g, w, JC, job, st, lane);
                st.V = 0; st.E = 0; st.L0 = 0; st.members = 0; st.cw_sum = 0; st.nov = 0;
            }
        }
#if POA_REPEAT == 4
        { const int no_ = st.nout; tile_emit(g, w, JC, job, st, lane); st.nout = no_; }
#endif
        tile_emit(g, w, JC, job, st, lane);
        if (lane == 0) { JC.out_n[job] = (uint32_t)st.nout | (edge ? 0x80000000u : 0u); if (ndrop && JC.dropped) atomicAdd(JC.dropped, ndrop); if (JC.stat_rows) atomicAdd(JC.stat_rows, nrows); }      // bit 31: a traceback touched a clipped band edge
        mem_sync();
    }
}

// One kernel per band width, so that each gets its own register budget: the 64-column instance keeps one DP cell per lane and fits
// POA_W1 waves per SIMD (its LDS working set is ~5.7 KB per tile); the wider ones need the 128 VGPRs of four waves per SIMD.
#ifndef POA_W1
#define POA_W1 6
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(POA_W1, POA_W1)))
void k_poa_tile1(PoaJobSet J, uint8_t* gscratch, size_t gbytes, uint32_t* __restrict__ work_ctr) { poa_tile_body<1>(J, gscratch, gbytes, work_ctr); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_poa_tile2(PoaJobSet J, uint8_t* gscratch, size_t gbytes, uint32_t* __restrict__ work_ctr) { poa_tile_body<2>(J, gscratch, gbytes, work_ctr); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_poa_tile4(PoaJobSet J, uint8_t* gscratch, size_t gbytes, uint32_t* __restrict__ work_ctr) { poa_tile_body<4>(J, gscratch, gbytes, work_ctr); }

// ------------------------------------------------------------------------------------------------ host side
int32_t poa_run_jobs(ngsid_ctx* ctx, PoaJobSet J, int band)
{
    if (J.njobs == 0) return NGSID_OK;
    if (!J.job_list) J.nrun = J.njobs;
    if (J.nrun == 0) return NGSID_OK;
    const int BW = band <= 64 ? 64 : (band <= 128 ? 128 : 256);
    if (J.g >= 0) NGSID_FAIL(ctx, NGSID_ERR_ARG, "POA gap score must be negative");
    if ((long long)J.m * J.Lmax >= 65536 || J.m < 0) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA local score range exceeds 16 bits (match %d x length %d)", J.m, J.Lmax);
    if (J.Vcap > 0xFFF0 || J.Ecap > 0xFFF0) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA graph capacity exceeds 16-bit indices (sequence too long for the tile engine)");
    const size_t lds = poa_lds_bytes(J.Vcap, J.Ecap, J.Lmax, BW);
    if (lds > 160 * 1024) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA tile needs %zu bytes of LDS (> 160 KiB): sequences too long", lds);
    const int wave_cap = BW == 64 ? 4 * POA_W1 : 16;                                  // waves per CU the register budget of the instance allows
    int per_cu = std::max<int>(1, std::min<int>(wave_cap, (int)((160 * 1024) / lds)));
    if (ngsid_opt(ctx, "poa_tiles_per_cu", 0) > 0) per_cu = std::max(1, std::min(per_cu, (int)ngsid_opt(ctx, "poa_tiles_per_cu", 0)));
    uint32_t nwg = (uint32_t)std::min<uint64_t>(J.nrun, (uint64_t)ctx->n_cu * per_cu);
    const size_t cells = (size_t)J.Vcap * BW;
    const size_t gbytes = poa_graph_bytes(J.Vcap, J.Ecap, J.Lmax);
    if (ctx->poa_h.n < nwg * cells) HIPCHK(ctx, ctx->poa_h.alloc(nwg * cells));
    if (ctx->poa_d.n < nwg * cells * 3 / 2) HIPCHK(ctx, ctx->poa_d.alloc(nwg * cells * 3 / 2));
    if (ctx->poa_g.n < nwg * gbytes) HIPCHK(ctx, ctx->poa_g.alloc(nwg * gbytes));
    J.Hglob = ctx->poa_h.p; J.dirglob = ctx->poa_d.p; J.covglob = nullptr; J.stat_rows = ctx->prof ? ctx->stat.p : nullptr;
    if (ctx->poa_ctr.n < 1) HIPCHK(ctx, ctx->poa_ctr.alloc(16));
    HIPCHK(ctx, hipMemsetAsync(ctx->poa_ctr.p, 0, sizeof(uint32_t), ctx->stream));
    ProfScope ps_(ctx, "k_poa_tile");
    if (BW == 64) { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_poa_tile1, dim3(nwg), dim3(64), lds, ctx->stream, J, ctx->poa_g.p, gbytes, ctx->poa_ctr.p); }
    else if (BW == 128) { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_poa_tile2, dim3(nwg), dim3(64), lds, ctx->stream, J, ctx->poa_g.p, gbytes, ctx->poa_ctr.p); }
    else { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_poa_tile4, dim3(nwg), dim3(64), lds, ctx->stream, J, ctx->poa_g.p, gbytes, ctx->poa_ctr.p); }
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}

// ---- device-driven hierarchy (poa_host.hip): one scratch allocation for the main band and the two wider redo instances, launches without host work
static int poa_per_cu(ngsid_ctx* ctx, int Vc, int Ec, int Lm, int BW)
{
    const size_t lds = poa_lds_bytes(Vc, Ec, Lm, BW);
    const int wave_cap = BW == 64 ? 4 * POA_W1 : 16;
    int per_cu = std::max<int>(1, std::min<int>(wave_cap, (int)((160 * 1024) / lds)));
    if (ngsid_opt(ctx, "poa_tiles_per_cu", 0) > 0) per_cu = std::max(1, std::min(per_cu, (int)ngsid_opt(ctx, "poa_tiles_per_cu", 0)));
    return per_cu;
}
int32_t poa_prepare(ngsid_ctx* ctx, PoaPlan& P, uint32_t max_jobs)
{
    const int B0 = P.band0 <= 64 ? 64 : (P.band0 <= 128 ? 128 : 256);
    P.band0 = B0;
    if (P.Vcap > 0xFFF0 || P.Ecap > 0xFFF0) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA graph capacity exceeds 16-bit indices (sequence too long for the tile engine)");
    const size_t gbytes = poa_graph_bytes(P.Vcap, P.Ecap, P.Lmax);
    // resident workgroups per CU; halved when the scratch does not fit (several contexts sharing one GPU, very long reads): the persistent
    // workgroups pull tiles from a queue, so fewer of them only lowers the parallelism
    for (int shrink = 1;; shrink *= 2) {
        size_t need_h = 0, need_d = 0, need_g = 0;
        for (int BW = B0; BW <= 256; BW *= 2) {
            const size_t lds = poa_lds_bytes(P.Vcap, P.Ecap, P.Lmax, BW);
            if (lds > 160 * 1024) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA tile needs %zu bytes of LDS (> 160 KiB): sequences too long", lds);
            uint32_t nwg;
            if (BW == B0) { nwg = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(max_jobs, 1u), std::max<uint64_t>(1, (uint64_t)ctx->n_cu * poa_per_cu(ctx, P.Vcap, P.Ecap, P.Lmax, BW) / shrink)); P.nwg_main = nwg; }
            else { nwg = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(max_jobs, 1u), std::max<uint64_t>(1, (uint64_t)ctx->n_cu * std::min(2, poa_per_cu(ctx, P.Vcap, P.Ecap, P.Lmax, BW)) / shrink)); P.nwg_redo = nwg; }       // redone tiles are rare: a small grid keeps the scratch small
            const size_t cells = (size_t)P.Vcap * BW;
            need_h = std::max(need_h, nwg * cells); need_d = std::max(need_d, nwg * cells * 3 / 2); need_g = std::max(need_g, nwg * gbytes);
        }
        if (B0 == 256) P.nwg_redo = 0;
        hipError_t e = hipSuccess;
        if (ctx->poa_h.n < need_h) e = ctx->poa_h.alloc(need_h);
        if (e == hipSuccess && ctx->poa_d.n < need_d) e = ctx->poa_d.alloc(need_d);
        if (e == hipSuccess && ctx->poa_g.n < need_g) e = ctx->poa_g.alloc(need_g);
        if (e == hipSuccess) break;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || shrink >= 64) HIPCHK(ctx, e);
    }
    return NGSID_OK;
}
int32_t poa_launch(ngsid_ctx* ctx, const PoaPlan& P, PoaJobSet J, int BW, bool redo, uint32_t* work_ctr)
{
    if (J.g >= 0) NGSID_FAIL(ctx, NGSID_ERR_ARG, "POA gap score must be negative");
    if ((long long)J.m * J.Lmax >= 65536 || J.m < 0) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA local score range exceeds 16 bits (match %d x length %d)", J.m, J.Lmax);
    const size_t lds = poa_lds_bytes(J.Vcap, J.Ecap, J.Lmax, BW);
    if (lds > 160 * 1024) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "POA tile needs %zu bytes of LDS (> 160 KiB): sequences too long", lds);      // (the caller checks all three instances before it starts)
    const uint32_t nwg = redo ? P.nwg_redo : P.nwg_main;
    if (nwg == 0) return NGSID_OK;
    const size_t gbytes = poa_graph_bytes(J.Vcap, J.Ecap, J.Lmax);
    J.Hglob = ctx->poa_h.p; J.dirglob = ctx->poa_d.p; J.covglob = nullptr; J.stat_rows = ctx->prof ? ctx->stat.p : nullptr;
    ProfScope ps_(ctx, redo ? "k_poa_tile_redo" : "k_poa_tile");
    if (BW == 64) { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_poa_tile1, dim3(nwg), dim3(64), lds, ctx->stream, J, ctx->poa_g.p, gbytes, work_ctr); }
    else if (BW == 128) { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_poa_tile2, dim3(nwg), dim3(64), lds, ctx->stream, J, ctx->poa_g.p, gbytes, work_ctr); }
    else { HIPCHK(ctx, hipFuncSetAttribute((const void*)k_poa_tile4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); hipLaunchKernelGGL(k_poa_tile4, dim3(nwg), dim3(64), lds, ctx->stream, J, ctx->poa_g.p, gbytes, work_ctr); }
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}
