// k_align16p.hip - the packed-int16 semi-global aligner with TWO PAIRS PER WAVE (round 3).
//
// Same DP, same tie-breaks, same outputs as k_sg_align16 / k_sg_align / the oracle; different packing.  k_sg_align16 puts two row halves of ONE pair
// in the two 16-bit halves of a register, one column apart, which costs a systolic skew of two columns per lane (m + 127 steps).  Here the low halves
// belong to pair 0 and the high halves to pair 1 of a work item: both work on the same column, the skew is one column per lane (max(m0, m1) + 63
// steps), a lane owns R consecutive query rows of each pair (single strip: n <= 64 R), and the per-step overhead (letters, masks, DPP hand-off, the
// store) is paid once for two pairs.  About 11 % fewer VALU instructions per pair in a loop that runs at the VALU issue rate.
// The capture of the last query row is a compile-time choice of the register (as in k_sg_align16), so the two pairs of an item must agree in
// (n - 1) mod R: the launcher bins the pairs of a length class by that residue (k_pair_bins / k_bin_offsets / k_bin_scatter) and an item is two
// consecutive entries of a bin (the last item of a bin may hold one pair).  Results are written by pair index, so the pairing does not show.
#include "ngsid_internal.h"
#include <algorithm>
#include <type_traits>

#define NEG16 (-20000)
// The traceback flags of this kernel are SIGN bits of packed 16-bit differences (d - h, E - mx, e_ext - E, f_ext - F): exact only while no difference wraps.  The
// largest magnitude is a cell at the "minus infinity" NEG16 (less one extension) against the largest score a pair of this kernel can reach - the bounds of
// ngsid_align16_applicable (match <= 4, ext <= 4, open <= 16) with queries of at most 896 bases (the single-strip classes): ADVICE r4.
static_assert(-(NEG16) + 4 /* ext */ + 16 /* open */ + 4 /* match */ * 896 /* query rows */ < 32768, "packed sign-bit flags of k_sg_align16p would wrap");
#define PKOP2(name, mnem) __device__ __forceinline__ int name(int a, int b) { int d; asm(mnem " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
PKOP2(pp_sub_i16, "v_pk_sub_i16")
PKOP2(pp_add_i16, "v_pk_add_i16")
PKOP2(pp_max_i16, "v_pk_max_i16")
PKOP2(pp_sub_u16, "v_pk_sub_u16")
__device__ __forceinline__ int pp_lshr16(int a, int k2_s) { int d; asm("v_pk_lshrrev_b16 %0, %1, %2" : "=v"(d) : "s"(k2_s), "v"(a)); return d; }  // per-half logical shift right; k2_s = the amount in BOTH halves of an SGPR (an inline constant would shift the high half by 0)
// (inline asm: from the C expression the compiler builds and / or3 chains of the same length as the min / shift / or form they replace; the masks live in SGPRs)
__device__ __forceinline__ int pp_bfi(int mask_s, int a, int b) { int d; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(mask_s), "v"(a), "v"(b)); return d; }        // (a & mask) | (b & ~mask)
__device__ __forceinline__ int pp_and_or(int a, int mask_s, int b) { int d; asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(mask_s), "v"(b)); return d; }  // (a & mask) | b
#define PKOP2S(name, mnem) __device__ __forceinline__ int name(int a, int b) { int d; asm(mnem " %0, %1, %2" : "=v"(d) : "v"(a), "s"(b)); return d; }
PKOP2S(pp_sub_i16_s, "v_pk_sub_i16")
PKOP2S(pp_min_u16_s, "v_pk_min_u16")
__device__ __forceinline__ int pp_mad_i16_sv(int a, int b_s, int c) { int d; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_s), "v"(c)); return d; }
__device__ __forceinline__ int pp_sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint8_t pp_perm_letter(uint8_t c) {          // same byte permutation as k_align16.hip: A,C,G,T -> 0..3, a,c,g,t -> 0x80..0x83
    const int b = ngsid_bcode(c);
    if (b < 4) return (uint8_t)(b | ((c & 0x20) ? 0x80 : 0));
    if ((c & 0x7C) == 0) { const int x = c & 3; const int up = x == 0 ? 'A' : x == 1 ? 'C' : x == 2 ? 'G' : 'T'; return (uint8_t)((c & 0x80) ? (up | 0x20) : up); }
    return c;
}
__device__ __forceinline__ int PP(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int PLO(int x) { return (int)(short)(x & 0xffff); }
__device__ __forceinline__ int PHI(int x) { return x >> 16; }
__device__ __forceinline__ int pp_sel(int mask, int a, int b) { return (a & mask) | (b & ~mask); }      // per bit: mask ? a : b (one v_bfi)

#define PBINS 32          // bins of one length class: residues (n - 1) mod R, R <= 16 (the rest unused)

template <int R>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4)))
void k_sg_align16p(AlignJob J, const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ bin_off /* [PBINS + 1] pairs */, const uint32_t* __restrict__ item_off /* [PBINS + 1] items */,
                   uint64_t* __restrict__ tb, uint64_t tb_words_per_wave /* per half */, uint32_t seq_lds, uint32_t* __restrict__ work_ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    uint8_t* tgt0 = smem; uint8_t* tgt1 = tgt0 + seq_lds; uint8_t* qry0 = tgt1 + seq_lds; uint8_t* qry1 = qry0 + seq_lds;
    uint64_t* tbblk = (uint64_t*)(smem + 4 * (size_t)seq_lds);            // traceback block: 64 steps x 8 lanes
    uint64_t* mytb0 = tb + (uint64_t)blockIdx.x * tb_words_per_wave * 2; uint64_t* mytb1 = mytb0 + tb_words_per_wave;
    constexpr int NA = (R + 3) / 4;                                       // flag accumulators: four nibbles (16 bits) per half each
    const uint32_t nitems = item_off[PBINS];

    for (;;) {
        uint32_t pq = 0; if (lane == 0) pq = atomicAdd(work_ctr, 1u);
        const uint32_t kq = (uint32_t)__builtin_amdgcn_readfirstlane((int)pq);
        if (kq >= nitems) break;
        int bin = 0;
        while (bin + 1 < PBINS && item_off[bin + 1] <= kq) ++bin;        // (<= 32 scalar loads; the items of a wave come in bin order)
        const uint32_t ib = kq - item_off[bin], b0 = bin_off[bin], cntb = bin_off[bin + 1] - b0;
        const bool have1 = 2 * ib + 1 < cntb;
        const uint64_t p0 = sorted[b0 + 2 * ib], p1 = have1 ? sorted[b0 + 2 * ib + 1] : p0;
        const uint32_t qi0 = J.qidx[p0], ti0 = J.tidx[p0], qi1 = J.qidx[p1], ti1 = J.tidx[p1];
        const uint8_t* q0 = J.qseq + J.qoff[qi0]; const uint8_t* t0 = J.tseq + J.toff[ti0];
        const uint8_t* q1 = J.qseq + J.qoff[qi1]; const uint8_t* t1 = J.tseq + J.toff[ti1];
        int n0 = pp_sgpr((int)(J.qoff[qi0 + 1] - J.qoff[qi0])), m0 = pp_sgpr((int)(J.toff[ti0 + 1] - J.toff[ti0]));
        int n1 = pp_sgpr((int)(J.qoff[qi1 + 1] - J.qoff[qi1])), m1 = pp_sgpr((int)(J.toff[ti1 + 1] - J.toff[ti1]));
        if (!have1) { n1 = 0; m1 = 0; }                                   // single pair: the high halves stay inactive
        // an empty target or an empty query (class 0 holds queries from 0 bases up): the outputs of k_sg_align16 for that case, the half takes no part in the DP
        auto degenerate = [&](uint64_t p, int n, int m) {
            if (lane == 0) {
                const int cols = n + m; const int mid = J.match_id ? J.match_id[p] : J.k;
                if (J.score) J.score[p] = 0; if (J.ncols) J.ncols[p] = cols; if (J.nmatch) J.nmatch[p] = 0;
                if (J.region) { int reg = (cols <= J.k) ? (0 >= mid) : ((0 >= mid) ? cols - J.k + 1 : 0); J.region[p] = reg; }
                if (J.span) { J.span[p * 4 + 0] = 0; J.span[p * 4 + 1] = 0; J.span[p * 4 + 2] = 0; J.span[p * 4 + 3] = 0; }
            }
            if (J.bp) for (int x = lane; x < J.bp_windows * 4; x += 64) J.bp[p * (uint64_t)J.bp_windows * 4 + x] = -1;
        };
        const bool deg0 = m0 <= 0 || n0 <= 0, deg1 = have1 && (m1 <= 0 || n1 <= 0);
        if (deg0) degenerate(p0, n0, m0);
        if (deg1) degenerate(p1, n1, m1);
        const int own_p = ((n0 > 0 ? n0 : 1) - 1) % R;                    // == (n1 - 1) % R: same bin
        if (deg0) { n0 = 0; m0 = 0; }
        if (deg1) { n1 = 0; m1 = 0; }
        if (deg0 && (deg1 || !have1)) continue;
        // letters outside ACGT (bits 0x7C of the permuted byte) score 0 against everything: two mask operations per cell that an item without such letters - nearly
        // every item - does not need (NOWILD instances of the step loop below; same results by construction)
        unsigned wild = 0;
        for (int x = lane; x < m0; x += 64) { const uint8_t v = pp_perm_letter(t0[x]); tgt0[x] = v; wild |= v; }
        for (int x = lane; x < n0; x += 64) { const uint8_t v = pp_perm_letter(q0[x]); qry0[x] = v; wild |= v; }
        for (int x = lane; x < m1; x += 64) { const uint8_t v = pp_perm_letter(t1[x]); tgt1[x] = v; wild |= v; }
        for (int x = lane; x < n1; x += 64) { const uint8_t v = pp_perm_letter(q1[x]); qry1[x] = v; wild |= v; }
        const bool nowild = __ballot((wild & 0x7Cu) != 0) == 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        const int OPEN2 = pp_sgpr(PP(J.open[p0], J.open[p1])), EXT2 = pp_sgpr(PP(J.ext, J.ext));
        const int MATCH2 = PP(J.match, J.match), NDIFF2 = pp_sgpr(PP(J.mismatch - J.match, J.mismatch - J.match));
        const int ONE2 = pp_sgpr(0x00010001);
        const int MK1 = pp_sgpr(0x20002000), MK2 = pp_sgpr(0x40004000), MK3 = pp_sgpr((int)0x80008000u), MKN = pp_sgpr((int)0xF000F000u);      // flag-nibble masks of the step
        const int SH2 = pp_sgpr(0x00020002), SH3 = pp_sgpr(0x00030003), SH4 = pp_sgpr(0x00040004);                                              // packed shift amounts
        const int mmax = m0 > m1 ? m0 : m1;
        const int steps = mmax + 63;
        const int own_lane0 = n0 > 0 ? (n0 - 1) / R : -1, own_lane1 = n1 > 0 ? (n1 - 1) / R : -1;
        int bestRowV0 = -(1 << 29), bestRowJ0 = 0, bestRowV1 = -(1 << 29), bestRowJ1 = 0;
        int hl2[R], e2[R], qc2[R], nwq2[R];
        const int i0 = lane * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = i0 + r;
            const int ca = i < n0 ? qry0[i] : 0x7C, cb = i < n1 ? qry1[i] : 0x7C;
            qc2[r] = PP(ca & 3, cb & 3); nwq2[r] = PP((ca & 0x7C) ? 0 : 0xffff, (cb & 0x7C) ? 0 : 0xffff);
            hl2[r] = 0; e2[r] = PP(NEG16, NEG16);
        }

        auto forward = [&](auto OPc, auto NWc) {
            constexpr int OWN_P = decltype(OPc)::value; constexpr bool NOWILD = decltype(NWc)::value;
            int hdiag2 = 0;                                               // H[i0-1][j-1], both pairs
            int send_h = 0, send_f = PP(NEG16, NEG16);                    // bottom row of this lane for the next one
            int keyRow0 = (int)0x80000000u, keyRow1 = (int)0x80000000u; const int jinv0 = 0xFFFF + lane;      // steady steps: packed (last-row value, -column) maxima
            // The steps in which EVERY lane stands inside both targets (tau - lane in [0, min(m0, m1)) for all 64 lanes: 63 <= tau < min(m0, m1), 84 % of the steps of a 750-base
            // pair) need none of the range masks: their own instance of the step (STEADY) drops the mask selects of the state (one per row, three per step) and the range tests.
            auto step = [&](int tau, auto STc) {
                constexpr bool STEADY = decltype(STc)::value;
                const int j = tau - lane;
                // lane 0 has no source lane: the shift leaves it the `old` operand - 0 for H (row -1 of the matrix), minus infinity for F (no select needed)
                const int hup = __builtin_amdgcn_update_dpp(0, send_h, 0x138, 0xf, 0xf, false), fup = __builtin_amdgcn_update_dpp(PP(NEG16, NEG16), send_f, 0x138, 0xf, 0xf, false);
                const bool a0 = STEADY || (j >= 0 && j < m0), a1 = STEADY || (j >= 0 && j < m1);      // STEADY: every lane stands inside both targets
                const int l0 = tgt0[STEADY ? j : (a0 ? j : 0)], l1 = tgt1[STEADY ? j : (a1 ? j : 0)];
                const int tc2 = PP(l0 & 3, l1 & 3);
                int nwt2 = 0; if constexpr (!NOWILD) nwt2 = PP((a0 && !(l0 & 0x7C)) ? 0xffff : 0, (a1 && !(l1 & 0x7C)) ? 0xffff : 0);
                const int am2 = STEADY ? -1 : PP(a0 ? 0xffff : 0, a1 ? 0xffff : 0);
                int hu2 = hup, f2 = fup, hd2 = hdiag2;
                int acc[NA], cap2 = 0;
#pragma unroll
                for (int a = 0; a < NA; ++a) acc[a] = 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e_ext = pp_sub_i16_s(e2[r], EXT2), e_opn = pp_sub_i16_s(hl2[r], OPEN2); const int E = pp_max_i16(e_ext, e_opn);
                    const int f_ext = pp_sub_i16_s(f2, EXT2), f_opn = pp_sub_i16_s(hu2, OPEN2); const int F = pp_max_i16(f_ext, f_opn);
                    const int z = pp_min_u16_s(qc2[r] ^ tc2, ONE2);                                // 1 = letters differ
                    int sc = pp_mad_i16_sv(z, NDIFF2, MATCH2);                                     // match / mismatch ...
                    if constexpr (!NOWILD) sc = sc & nwq2[r] & nwt2;                                // ... / 0 for wildcards (and for columns outside the target, whose cells nothing reads)
                    const int d = pp_add_i16(hd2, sc);
                    const int mx = pp_max_i16(E, F); const int h = pp_max_i16(d, mx);
                    // COMPLEMENT flags (1 = "not equal"): bit0 h!=d, bit1 mx!=E (F>E), bit2 E!=e_ext (opened), bit3 F!=f_ext (opened)
                    // each flag is the SIGN of a packed difference (smaller - larger: negative iff they differ); the four signs are moved to bits 12..15 of their half by
                    // per-half shifts and merged with bit-field inserts (6 operations instead of 4 min + 3 shift + ors), and the nibble enters its accumulator at the TOP:
                    // the k-th of the cnt rows of an accumulator ends in nibble (4 - cnt + k) of its half (round 4)
                    const int s0 = pp_sub_i16(d, h), s1 = pp_sub_i16(E, mx), s2 = pp_sub_i16(e_ext, E), s3 = pp_sub_i16(f_ext, F);
                    int c = pp_bfi(MK1, pp_lshr16(s1, SH2), pp_lshr16(s0, SH3));
                    c = pp_bfi(MK2, pp_lshr16(s2, ONE2), c);
                    c = pp_bfi(MK3, s3, c);
                    acc[r >> 2] = pp_and_or(c, MKN, pp_lshr16(acc[r >> 2], SH4));                   // four nibbles per half and accumulator
                    hd2 = hl2[r];
                    hl2[r] = STEADY ? h : pp_sel(am2, h, hl2[r]);
                    e2[r] = E;                 // not masked (see k_align16.hip: before a lane's first column E only relaxes to -open, after its last it is not used)
                    hu2 = h; f2 = F;
                    if (r == OWN_P) cap2 = h;
                }
                // traceback words: pair 0 = the low halves of the accumulators, pair 1 = the high halves (complement nibbles); accumulator a sits at bits [16 a, 16 a + 16)
                unsigned long long w0 = 0, w1 = 0;
#pragma unroll
                for (int a = 0; a < NA; ++a) { w0 |= (unsigned long long)((unsigned)acc[a] & 0xffffu) << (16 * a); w1 |= (unsigned long long)((unsigned)acc[a] >> 16) << (16 * a); }
                mytb0[(uint64_t)tau * 64 + lane] = w0;
                mytb1[(uint64_t)tau * 64 + lane] = w1;
                if (STEADY) { hdiag2 = hup; send_h = hu2; send_f = f2; }
                else { hdiag2 = pp_sel(am2, hup, hdiag2); send_h = pp_sel(am2, hu2, send_h); send_f = pp_sel(am2, f2, send_f); }
                if (STEADY) {
                    // last query row, first maximum over the columns, as ONE signed maximum per pair of (value << 16 | 0xFFFF - column): equal values keep the larger low
                    // half = the smaller column.  Every lane runs it, only the lane that owns the last row is read (merged into bestRow* after the steady steps).
                    const int jinv = jinv0 - tau;
                    keyRow0 = max(keyRow0, (int)(((unsigned)cap2 << 16) | (unsigned)jinv));
                    keyRow1 = max(keyRow1, (int)(((unsigned)cap2 & 0xFFFF0000u) | (unsigned)jinv));
                } else {   // last query row: first maximum over the columns
                    const int v0 = PLO(cap2), v1 = PHI(cap2);
                    const bool b0_ = a0 && lane == own_lane0 && v0 > bestRowV0; bestRowV0 = b0_ ? v0 : bestRowV0; bestRowJ0 = b0_ ? j : bestRowJ0;
                    const bool b1_ = a1 && lane == own_lane1 && v1 > bestRowV1; bestRowV1 = b1_ ? v1 : bestRowV1; bestRowJ1 = b1_ ? j : bestRowJ1;
                }
            };
            const int mmin = m0 < m1 ? m0 : m1;
            const int st_lo = 63 < steps ? 63 : steps, st_hi = mmin > st_lo ? (mmin < steps ? mmin : steps) : st_lo;
            int tau = 0;
            for (; tau < st_lo; ++tau) step(tau, std::false_type{});
            // two steps per iteration: the state a step leaves (one register per row: the new H of the row, while the old one is still the diagonal input of the row below)
            // is consumed by the second step in place, instead of being copied back into the loop's registers after every step (~38 v_mov per step of ~340 instructions)
            for (; tau + 1 < st_hi; tau += 2) { step(tau, std::true_type{}); step(tau + 1, std::true_type{}); }
            for (; tau < st_hi; ++tau) step(tau, std::true_type{});
            if (st_hi > st_lo) {      // the steady steps' maxima join the running ones (their columns lie between those of the ramp-up and of the ramp-down steps)
                const int v0 = keyRow0 >> 16, v1 = keyRow1 >> 16;
                if (lane == own_lane0 && v0 > bestRowV0) { bestRowV0 = v0; bestRowJ0 = 0xFFFF - (keyRow0 & 0xFFFF); }
                if (lane == own_lane1 && v1 > bestRowV1) { bestRowV1 = v1; bestRowJ1 = 0xFFFF - (keyRow1 & 0xFFFF); }
            }
            for (; tau < steps; ++tau) step(tau, std::false_type{});
        };
        switch (own_p) {
#define PCASE(k) case k: if (nowild) forward(std::integral_constant<int, (R > k ? k : 0)>{}, std::true_type{}); else forward(std::integral_constant<int, (R > k ? k : 0)>{}, std::false_type{}); break;
            PCASE(0) PCASE(1) PCASE(2) PCASE(3) PCASE(4) PCASE(5) PCASE(6) PCASE(7) PCASE(8) PCASE(9) PCASE(10) PCASE(11) PCASE(12) PCASE(13) PCASE(14)
#undef PCASE
            default: if (nowild) forward(std::integral_constant<int, (R > 15 ? 15 : 0)>{}, std::true_type{}); else forward(std::integral_constant<int, (R > 15 ? 15 : 0)>{}, std::false_type{}); break;
        }
        // last target column of each pair: the state holds H[i][m-1] for every row
        int bestColV0 = -(1 << 29), bestColI0 = 0x7fffffff, bestColV1 = -(1 << 29), bestColI1 = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = i0 + r; const int v0 = PLO(hl2[r]), v1 = PHI(hl2[r]);
            if (i < n0 && v0 > bestColV0) { bestColV0 = v0; bestColI0 = i; }
            if (i < n1 && v1 > bestColV1) { bestColV1 = v1; bestColI1 = i; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        const AlignJob* Jt = (const AlignJob*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(Jt));
        for (int half = 0; half < 2; ++half) {
            const uint64_t p = half ? p1 : p0; const int n = half ? n1 : n0, m = half ? m1 : m0;
            if (n <= 0) continue;                          // no second pair / degenerate pair (handled above)
            const uint8_t* qry = half ? qry1 : qry0; const uint8_t* tgt = half ? tgt1 : tgt0;
            const uint64_t* mytb = half ? mytb1 : mytb0;
            // ---- end cell: first maximum over the last row, then strictly larger over the last column with the lowest row
            int rowV = half ? bestRowV1 : bestRowV0, rowJ = half ? bestRowJ1 : bestRowJ0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { int ov = __shfl_xor(rowV, d), oj = __shfl_xor(rowJ, d); if (ov > rowV) { rowV = ov; rowJ = oj; } }
            int colV = half ? bestColV1 : bestColV0, colI = half ? bestColI1 : bestColI0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { int ov = __shfl_xor(colV, d), oi = __shfl_xor(colI, d); if (ov > colV || (ov == colV && oi < colI)) { colV = ov; colI = oi; } }
            int ei = n - 1, ej = rowJ, best = rowV;
            if (colV > best) { best = colV; ei = colI; ej = m - 1; }
            ei = pp_sgpr(ei); ej = pp_sgpr(ej); best = pp_sgpr(best);

            // ---- traceback (uniform over the wave): the bookkeeping of k_sg_align16, the word of cell (i, j) is at step j + i / R, lane i / R, nibble i % R
            if (Jt->bp) for (int x = lane; x < Jt->bp_windows * 4; x += 64) Jt->bp[p * (uint64_t)Jt->bp_windows * 4 + x] = -1;
            const int K = Jt->k; const int mid = Jt->match_id ? Jt->match_id[p] : K;
            const uint64_t kmask = (K >= 64) ? ~0ull : ((1ull << K) - 1);
            uint64_t win = 0; int cols = 0, nm = 0, region = 0;
            {
                const int z = (n - 1 - ei) + (m - 1 - ej);
                const int zl = z < K ? z : K;
                for (int x = 0; x < zl; ++x) { win <<= 1; ++cols; if (cols >= K) region += ((int)__popcll(win & kmask) >= mid); }
                if (z > zl) { region += (0 >= mid) ? (z - zl) : 0; cols += z - zl; }
            }
            int i = ei, j = ej, state = 0;
            int q_end = -1, t_end = -1, q_beg = -1, t_beg = -1;
            int cw = -1, w_qf = 0, w_ql = 0, w_tf = 0, w_tl = 0;
            int32_t* bpp = Jt->bp ? Jt->bp + p * (uint64_t)Jt->bp_windows * 4 : nullptr;
            int blk_g = -1, blk_hi = -1;
            int wsn = bpp ? j / Jt->window : 0, ws = bpp ? wsn * Jt->window : 0;
            // (every round of the walk takes at least one step or reloads a block once per 64 steps: the bound is never reached; it turns a corrupted traceback word into a wrong
            // result the parity tests catch instead of a wave that never ends)
            for (int guard = 4 * (n + m) + 512; i >= 0 && j >= 0 && guard > 0; --guard) {
                if (bpp) while (j < ws) { ws -= Jt->window; --wsn; }
                {   // the block of traceback words around the current cell in LDS (64 steps x one group of 8 lanes)
                    const int l = i / R; const int tau = j + l; const int grp = l >> 3;
                    if (grp != blk_g || tau > blk_hi || tau < blk_hi - 63) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        blk_g = grp; blk_hi = tau;
                        const int tt = tau - lane;
                        if (tt >= 0) {
                            const uint4* src = (const uint4*)(mytb + (uint64_t)tt * 64 + grp * 8);
                            ngsid_v4u* dstp = (ngsid_v4u*)(tbblk + lane * 8);
                            dstp[0] = ngsid_load16_l2(src + 0); dstp[1] = ngsid_load16_l2(src + 1); dstp[2] = ngsid_load16_l2(src + 2); dstp[3] = ngsid_load16_l2(src + 3);
                        }
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                // lane k decodes the cell k diagonal steps back; in state 0 the wave takes the whole leading run of diagonal moves at once
                const int ik = i - lane, jk = j - lane;
                bool inb = false; int vk = 0;
                if (ik >= 0 && jk >= 0) {
                    const int l = ik / R; const int r = ik - l * R;
                    const int tau = jk + l;
                    if ((l >> 3) == blk_g && tau <= blk_hi && tau >= blk_hi - 63) {
                        const uint64_t word = tbblk[(blk_hi - tau) * 8 + (l & 7)];
                        const int a = r >> 2; const int cnt_a = (R - 4 * a) < 4 ? (R - 4 * a) : 4;
                        const int sh = 16 * a + 4 * (4 - cnt_a + (r & 3));
                        vk = (int)((~(word >> sh)) & 15);        // stored complemented -> bit0 diag, bit1 E>=F, bit2 E extends, bit3 F extends
                        inb = true;
                    }
                }
                int run = 0;
                if (state == 0) {
                    const bool good = inb && (vk & 1) && jk >= ws;          // a run never crosses a polishing-window boundary
                    const unsigned long long gm = __ballot(good);
                    run = (~gm) ? __builtin_ctzll(~gm) : 64;
                }
                if (run > 0) {
                    const unsigned long long mb = __ballot(ik >= 0 && jk >= 0 && qry[ik >= 0 ? ik : 0] == tgt[jk >= 0 ? jk : 0]);      // match bit of step k
                    const unsigned long long rmask = run == 64 ? ~0ull : ((1ull << run) - 1);
                    const uint64_t wk = (lane == 63 ? 0ull : (win << (lane + 1))) | (__brevll(mb) >> (63 - lane));
                    const bool cnt = (cols + lane + 1 >= K) && ((int)__popcll(wk & kmask) >= mid);
                    region += (int)__popcll(__ballot(cnt) & rmask);
                    nm += (int)__popcll(mb & rmask);
                    { const int last = run - 1; const unsigned lo_ = __builtin_amdgcn_readlane((unsigned)wk, last), hi_ = __builtin_amdgcn_readlane((unsigned)(wk >> 32), last); win = ((uint64_t)hi_ << 32) | lo_; }
                    cols += run;
                    if (q_end < 0) { q_end = i; t_end = j; }
                    q_beg = i - run + 1; t_beg = j - run + 1;
                    if (bpp) {
                        const int wn = wsn;
                        if (wn != cw) { if (lane == 0 && cw >= 0 && cw < Jt->bp_windows) { bpp[cw * 4 + 0] = w_qf; bpp[cw * 4 + 1] = w_ql; bpp[cw * 4 + 2] = w_tf; bpp[cw * 4 + 3] = w_tl; } cw = wn; w_ql = i; w_tl = j; }
                        w_qf = i - run + 1; w_tf = j - run + 1;
                    }
                    i -= run; j -= run;
                    if (i < 0 || j < 0) break;
                    if (bpp) while (j < ws) { ws -= Jt->window; --wsn; }
                }
                if (run == 64 || !((__ballot(inb) >> run) & 1)) continue;     // next cell outside the loaded block: go round (reloads)
                const int v = __builtin_amdgcn_readlane(vk, run);
                int bit = 0, emit = 1;
                if (state == 0) {
                    if (v & 1) {                                       // (a diagonal move the run could not take: window boundary)
                        bit = (qry[i] == tgt[j]);
                        if (q_end < 0) { q_end = i; t_end = j; }
                        q_beg = i; t_beg = j;
                        if (bpp) {
                            const int wn = wsn;
                            if (wn != cw) { if (lane == 0 && cw >= 0 && cw < Jt->bp_windows) { bpp[cw * 4 + 0] = w_qf; bpp[cw * 4 + 1] = w_ql; bpp[cw * 4 + 2] = w_tf; bpp[cw * 4 + 3] = w_tl; } cw = wn; w_ql = i; w_tl = j; }
                            w_qf = i; w_tf = j;
                        }
                        --i; --j;
                    } else { state = (v & 2) ? 1 : 2; emit = 0; }
                } else if (state == 1) { if (!((v >> 2) & 1)) state = 0; --j; }
                else { if (!((v >> 3) & 1)) state = 0; --i; }
                if (emit) { win = (win << 1) | (uint64_t)bit; nm += bit; ++cols; if (cols >= K) region += ((int)__popcll(win & kmask) >= mid); }
            }
            if (lane == 0 && bpp && cw >= 0 && cw < Jt->bp_windows) { bpp[cw * 4 + 0] = w_qf; bpp[cw * 4 + 1] = w_ql; bpp[cw * 4 + 2] = w_tf; bpp[cw * 4 + 3] = w_tl; }
            {
                const int z = (i + 1) + (j + 1);
                const int zl = z < K ? z : K;
                for (int x = 0; x < zl; ++x) { win <<= 1; ++cols; if (cols >= K) region += ((int)__popcll(win & kmask) >= mid); }
                if (z > zl) { region += (0 >= mid) ? (z - zl) : 0; cols += z - zl; }
            }
            if (cols < K) region = (nm >= mid) ? 1 : 0;
            if (lane == 0) {
                if (Jt->score) Jt->score[p] = best;
                if (Jt->ncols) Jt->ncols[p] = cols;
                if (Jt->nmatch) Jt->nmatch[p] = nm;
                if (Jt->region) Jt->region[p] = region;
                if (Jt->span) { Jt->span[p * 4 + 0] = q_beg; Jt->span[p * 4 + 1] = q_end; Jt->span[p * 4 + 2] = t_beg; Jt->span[p * 4 + 3] = t_end; }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- binning of one length class by (n - 1) mod R.  `list` / `count` = the class list of k_pair_classes (queries of 513 - 896 bases: n >= 1).
__global__ __launch_bounds__(256)
void k_pair_bins(AlignJob J, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, int R, uint8_t* __restrict__ bin_of, uint32_t* __restrict__ bin_cnt /* [PBINS] */)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int bin = -1;
    if (k < *count) {
        const uint64_t p = list[k]; const uint32_t qi = J.qidx[p];
        const int n = (int)(J.qoff[qi + 1] - J.qoff[qi]);
        bin = n >= 1 ? (n - 1) % R : 0;
        bin_of[k] = (uint8_t)bin;
    }
    for (int b = 0; b < PBINS; ++b) {                     // one atomic per wave and bin
        const unsigned long long mk = __ballot(bin == b);
        if (mk && lane == (int)__builtin_ctzll(mk)) atomicAdd(&bin_cnt[b], (uint32_t)__popcll(mk));
    }
}
__global__ void k_bin_offsets(const uint32_t* __restrict__ bin_cnt, uint32_t* __restrict__ bin_off, uint32_t* __restrict__ item_off, uint32_t* __restrict__ cursor)
{
    if (threadIdx.x != 0) return;
    uint32_t a = 0, it = 0;
    for (int b = 0; b < PBINS; ++b) { bin_off[b] = a; item_off[b] = it; cursor[b] = 0; a += bin_cnt[b]; it += (bin_cnt[b] + 1) / 2; }
    bin_off[PBINS] = a; item_off[PBINS] = it;
}
__global__ __launch_bounds__(256)
void k_bin_scatter(const uint32_t* __restrict__ list, const uint32_t* __restrict__ count, const uint8_t* __restrict__ bin_of, const uint32_t* __restrict__ bin_off, uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int bin = k < *count ? (int)bin_of[k] : -1;
    for (int b = 0; b < PBINS; ++b) {
        const unsigned long long mk = __ballot(bin == b);
        if (!mk) continue;
        const int leader = (int)__builtin_ctzll(mk);
        uint32_t base = 0; if (lane == leader) base = atomicAdd(&cursor[b], (uint32_t)__popcll(mk));
        base = __shfl(base, leader);
        if (bin == b) sorted[bin_off[b] + base + __popcll(mk & ((1ull << lane) - 1))] = list[k];
    }
}

struct LaunchP { uint64_t words_half, nwaves; uint32_t seq_lds; size_t lds; };
template <int R>
static int32_t plan16p(ngsid_ctx* ctx, uint64_t npairs, uint32_t max_tlen, LaunchP* L)
{
    L->seq_lds = (std::max<uint32_t>(max_tlen, 64u * R) + 15u) & ~15u;
    L->lds = 4 * (size_t)L->seq_lds + 4096;
    int occ = 0;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sg_align16p<R>, 64, L->lds));
    if (occ < 1) occ = 1;
    L->words_half = ((uint64_t)max_tlen + 63) * 64;
    uint64_t want = std::min<uint64_t>((npairs + 1) / 2, (uint64_t)occ * ctx->n_cu);
    const uint64_t by_mem = ctx->scratch_budget / (2 * L->words_half * 8 + 1);
    L->nwaves = std::max<uint64_t>(1, std::min(want, by_mem));
    return NGSID_OK;
}
// traceback words (u64) the paired launch of length class `cls` (0: <= 256 bases, R = 4 rows per lane; 1: <= 512, R = 8; 2: <= 768, R = 12; 3: <= 896, R = 14) needs for a batch of npairs
int32_t ngsid_paired_tb_words(ngsid_ctx* ctx, int cls, uint64_t npairs, uint32_t max_tlen, uint64_t* words)
{
    LaunchP L; int32_t rc = cls == 0 ? plan16p<4>(ctx, npairs, max_tlen, &L) : cls == 1 ? plan16p<8>(ctx, npairs, max_tlen, &L) : cls == 2 ? plan16p<12>(ctx, npairs, max_tlen, &L) : plan16p<14>(ctx, npairs, max_tlen, &L);
    if (rc) return rc;
    *words = L.nwaves * 2 * L.words_half;
    return NGSID_OK;
}
template <int R>
static int32_t launch_paired(ngsid_ctx* ctx, const AlignJob& job, int cls, uint32_t max_tlen, hipStream_t st, uint32_t* ibase, uint32_t* sorted, uint8_t* bin_of, uint64_t* tb)
{
    const uint64_t n = job.npairs;
    const uint32_t* list = ctx->aln_cls.p + (size_t)cls * n; const uint32_t* count = ctx->aln_ctr.p + 8 + cls;
    uint32_t* bin_cnt = ibase, *bin_off = ibase + PBINS, *item_off = bin_off + PBINS + 1, *cursor = item_off + PBINS + 1, *wctr = cursor + PBINS;
    HIPCHK(ctx, hipMemsetAsync(ibase, 0, (size_t)(4 * PBINS + 8) * sizeof(uint32_t), st));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_pair_bins, dim3(nb), dim3(256), 0, st, job, list, count, R, bin_of, bin_cnt);
    hipLaunchKernelGGL(k_bin_offsets, dim3(1), dim3(64), 0, st, (const uint32_t*)bin_cnt, bin_off, item_off, cursor);
    hipLaunchKernelGGL(k_bin_scatter, dim3(nb), dim3(256), 0, st, list, count, (const uint8_t*)bin_of, (const uint32_t*)bin_off, cursor, sorted);
    HIPCHK(ctx, hipGetLastError());
    LaunchP L; int32_t rc = plan16p<R>(ctx, n, max_tlen, &L); if (rc) return rc;
    HIPCHK(ctx, hipFuncSetAttribute((const void*)k_sg_align16p<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds));
    { ProfScope ps_(ctx, st == ctx->stream ? "k_sg_align" : "k_sg_align_side", st);
      hipLaunchKernelGGL((k_sg_align16p<R>), dim3((unsigned)L.nwaves), dim3(64), L.lds, st, job, (const uint32_t*)sorted, (const uint32_t*)bin_off, (const uint32_t*)item_off, tb, L.words_half, L.seq_lds, wctr); }
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}
// One length class (0 - 3) of a partitioned batch (class lists of ngsid_partition_pairs in ctx->aln_cls / aln_ctr) through the paired kernel on stream st;
// tb = this launch's slice of the traceback scratch (ngsid_paired_tb_words).
int32_t ngsid_launch_paired_class(ngsid_ctx* ctx, const AlignJob& job, int cls, uint32_t max_tlen, hipStream_t st, uint64_t* tb)
{
    const uint64_t n = job.npairs;
    const int slot = cls;                                  // the class launches of a call run concurrently: one slice of the scratch each
    const size_t ints = 4 * PBINS + 8;
    if (ctx->aln_pint.n < 4 * ints) HIPCHK(ctx, ctx->aln_pint.alloc(4 * ints));
    if (ctx->aln_psorted.n < 4 * n) HIPCHK(ctx, ctx->aln_psorted.reserve(4 * n));
    if (ctx->aln_pbin.n < 4 * n) HIPCHK(ctx, ctx->aln_pbin.reserve(4 * n));
    uint32_t* ibase = ctx->aln_pint.p + slot * ints; uint32_t* sorted = ctx->aln_psorted.p + (size_t)slot * n; uint8_t* bin_of = ctx->aln_pbin.p + (size_t)slot * n;
    return cls == 0 ? launch_paired<4>(ctx, job, cls, max_tlen, st, ibase, sorted, bin_of, tb) : cls == 1 ? launch_paired<8>(ctx, job, cls, max_tlen, st, ibase, sorted, bin_of, tb)
         : cls == 2 ? launch_paired<12>(ctx, job, cls, max_tlen, st, ibase, sorted, bin_of, tb) : launch_paired<14>(ctx, job, cls, max_tlen, st, ibase, sorted, bin_of, tb);
}
