// k_align16.hip - packed-int16 variant of the semi-global affine aligner (k_align.hip) for sequences <= 4000 bases.
//
// Same DP, same tie-breaks, same outputs (bit-identical to k_sg_align and to the oracle); different mapping:
// a lane owns 2*RP consecutive query rows split in two halves A (rows i0..i0+RP-1) and B (rows i0+RP..i0+2RP-1).
// Register p holds row p of A in its low 16 bits and row p of B in its high 16 bits, and the two halves work one
// column apart (A on column j, B on column j-1), which makes the two 16-bit lanes of every register independent
// cells: all max/add/sub are v_pk_*_i16 (two cells per VALU op).  Traceback flags are derived without compares
// (flag = 1 - min(u16(max - candidate), 1)) and collected in packed accumulators; per step a lane still emits one
// 8-byte word (4 bits per cell, A cells in the low dword, B cells in the high dword).
// The systolic skew is two columns per lane: steps = m + 127 per strip.
#include "ngsid_internal.h"
#include <algorithm>
#include <type_traits>

#define NEG16 (-20000)
// Packed 16-bit VALU ops through inline asm: with plain vector types the compiler "simplifies" the flag arithmetic back into
// per-half compares + selects (no packed compare exists), which costs more than the 32-bit kernel.
#define PKOP2(name, mnem) __device__ __forceinline__ int name(int a, int b) { int d; asm(mnem " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
PKOP2(pk_sub_i16, "v_pk_sub_i16")
PKOP2(pk_add_i16, "v_pk_add_i16")
PKOP2(pk_max_i16, "v_pk_max_i16")
PKOP2(pk_sub_u16, "v_pk_sub_u16")
// second operand wave-uniform (lives in an SGPR: one constant-bus read per instruction is allowed on gfx9)
#define PKOP2S(name, mnem) __device__ __forceinline__ int name(int a, int b) { int d; asm(mnem " %0, %1, %2" : "=v"(d) : "v"(a), "s"(b)); return d; }
PKOP2S(pk_sub_i16_s, "v_pk_sub_i16")
PKOP2S(pk_min_u16_s, "v_pk_min_u16")
__device__ __forceinline__ int pk_mad_i16_sv(int a, int b_s, int c) { int d; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(b_s), "v"(c)); return d; }
__device__ __forceinline__ int sgpr(int x) { return __builtin_amdgcn_readfirstlane(x); }
// Staged letters are stored through a byte PERMUTATION that sends A,C,G,T to 0..3 and a,c,g,t to 0x80..0x83 (and those eight
// byte values back to the letters), so raw-character equality is preserved and the DP loop decodes with two ANDs.
__device__ __forceinline__ uint8_t perm_letter(uint8_t c) {
    const int b = ngsid_bcode(c);
    if (b < 4) return (uint8_t)(b | ((c & 0x20) ? 0x80 : 0));
    if ((c & 0x7C) == 0) { const int x = c & 3; const int up = x == 0 ? 'A' : x == 1 ? 'C' : x == 2 ? 'G' : 'T'; return (uint8_t)((c & 0x80) ? (up | 0x20) : up); }
    return c;
}
__device__ __forceinline__ int PK(int lo, int hi) { return (lo & 0xffff) | (hi << 16); }
__device__ __forceinline__ int LO16(int x) { return (int)(short)(x & 0xffff); }
__device__ __forceinline__ int HI16(int x) { return x >> 16; }

template <int RP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8)))
void k_sg_align16(AlignJob J, uint64_t* __restrict__ tb, uint64_t tb_words_per_wave, int32_t* __restrict__ bnd, uint32_t bnd_stride, uint32_t lds_per_wave, uint32_t* __restrict__ work_ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    uint8_t* tgt = smem + (size_t)wib * lds_per_wave;
    const uint32_t seq_lds = (lds_per_wave - 4096) / 2;
    uint8_t* qry = tgt + seq_lds;
    uint64_t* tbblk = (uint64_t*)(tgt + 2 * (size_t)seq_lds);
    uint64_t* mytb = tb + wave * tb_words_per_wave;
    int32_t* mybnd = bnd + wave * (uint64_t)bnd_stride * 2;
    constexpr int RPL = 2 * RP;
    constexpr int STRIP = 64 * RPL;
    constexpr int C0 = RP < 4 ? RP : 4;            // pairs collected in accumulator 0
    constexpr int C1 = RP - C0;                     // pairs collected in accumulator 1

    for (;;) {
        // persistent waves pull pairs from a queue: the grid is sized to what is resident, so there is no tail of idle SIMDs
        uint32_t pq = 0; if (lane == 0) pq = atomicAdd(work_ctr, 1u);
        const uint32_t kq = (uint32_t)__builtin_amdgcn_readfirstlane((int)pq);
        if (kq >= (J.npairs_dev ? *J.npairs_dev : (uint32_t)J.npairs)) break;
        const uint64_t p = J.pair_list ? J.pair_list[kq] : kq;
        const uint32_t qi = J.qidx[p], ti = J.tidx[p];
        const uint8_t* q = J.qseq + J.qoff[qi]; const int n = sgpr((int)(J.qoff[qi + 1] - J.qoff[qi]));     // wave-uniform by construction
        const uint8_t* t = J.tseq + J.toff[ti]; const int m = sgpr((int)(J.toff[ti + 1] - J.toff[ti]));
        if (n <= 0 || m <= 0) {
            if (lane == 0) {
                const int cols = n + m; const int mid = J.match_id ? J.match_id[p] : J.k;
                if (J.score) J.score[p] = 0; if (J.ncols) J.ncols[p] = cols; if (J.nmatch) J.nmatch[p] = 0;
                if (J.region) { int reg = (cols <= J.k) ? (0 >= mid) : ((0 >= mid) ? cols - J.k + 1 : 0); J.region[p] = reg; }
                if (J.span) { J.span[p * 4 + 0] = 0; J.span[p * 4 + 1] = 0; J.span[p * 4 + 2] = 0; J.span[p * 4 + 3] = 0; }
            }
            if (J.bp) for (int x = lane; x < J.bp_windows * 4; x += 64) J.bp[p * (uint64_t)J.bp_windows * 4 + x] = -1;
            continue;
        }
        for (int x = lane; x < m; x += 64) tgt[x] = perm_letter(t[x]);
        for (int x = lane; x < n; x += 64) qry[x] = perm_letter(q[x]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        const int OPEN2 = sgpr(PK(J.open[p], J.open[p])), EXT2 = sgpr(PK(J.ext, J.ext));
        const int MATCH2 = PK(J.match, J.match), NDIFF2 = sgpr(PK(J.mismatch - J.match, J.mismatch - J.match));
        const int ONE2 = sgpr(0x00010001);
        const int steps = m + 127;
        const int nstrips = (n + STRIP - 1) / STRIP;
        // owner of the last query row (wave-uniform)
        const int own_il = (n - 1) % STRIP, own_lane = own_il / RPL, own_rr = own_il % RPL, own_half = own_rr / RP, own_p = own_rr % RP;
        int bestRowV = -(1 << 29), bestRowJ = 0;
        int bestColV = -(1 << 29), bestColI = 0x7fffffff;

        // The strip loop is instantiated once per register pair that can hold the LAST query row (wave-uniform own_p): the capture of that row's
        // cell is then a compile-time choice instead of RP selects per step.
        auto strips = [&](auto OPc) {
        constexpr int OWN_P = decltype(OPc)::value;
        for (int sidx = 0; sidx < nstrips; ++sidx) {
            const int i0 = sidx * STRIP + lane * RPL;
            int qc2[RP], nwq2[RP], hl2[RP], e2[RP];
#pragma unroll
            for (int r = 0; r < RP; ++r) {
                const int ia = i0 + r, ib = i0 + RP + r;
                const int ca = ia < n ? qry[ia] : 0x7C, cb = ib < n ? qry[ib] : 0x7C;
                qc2[r] = PK(ca & 3, cb & 3); nwq2[r] = PK((ca & 0x7C) ? 0 : 0xffff, (cb & 0x7C) ? 0 : 0xffff);
                hl2[r] = 0; e2[r] = PK(NEG16, NEG16);
            }
            int hdiagA = 0;                              // H[i0-1][jA-1]
            int aBot_h1 = 0, aBot_h2 = 0, aBot_f1 = NEG16; // A's bottom row at columns jA-1 (H,F) and jA-2 (H)
            int send_h = 0, send_f = NEG16;               // B's bottom row for the next lane
            const bool last_strip = sidx + 1 == nstrips;
            uint64_t* stb = mytb + (uint64_t)sidx * steps * 64;
            int tc2 = 0, nwt2 = 0, am2 = 0;              // target letter / not-wildcard mask / active mask: low half = A now, high half = A one step ago = B now
            for (int tau = 0; tau < steps; ++tau) {
                const int jA = tau - 2 * lane, jB = jA - 1;
                int hupA = __builtin_amdgcn_update_dpp(0, send_h, 0x138, 0xf, 0xf, false), fupA = __builtin_amdgcn_update_dpp(0, send_f, 0x138, 0xf, 0xf, false);
                const bool actA = jA >= 0 && jA < m;
                const bool actB = am2 & 1;                // A was active one step ago <=> column jB is inside the matrix
                if (lane == 0) {
                    if (sidx == 0) { hupA = 0; fupA = NEG16; }
                    else if (actA) { hupA = __hip_atomic_load(&mybnd[jA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); fupA = __hip_atomic_load(&mybnd[bnd_stride + jA], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                }
                // no early exit and no branches on activity: inactive halves are masked out of the state updates
                const int pA = tgt[actA ? jA : 0];
                tc2 = (tc2 << 16) | (pA & 3);
                nwt2 = (nwt2 << 16) | ((actA && !(pA & 0x7C)) ? 0xffff : 0);
                am2 = (am2 << 16) | (actA ? 0xffff : 0);
                int hu2 = PK(hupA, aBot_h1), f2 = PK(fupA, aBot_f1), hd2 = PK(hdiagA, aBot_h2);
                int acc0 = 0, acc1 = 0, cap2 = 0;
#pragma unroll
                for (int r = 0; r < RP; ++r) {
                    const int e_ext = pk_sub_i16_s(e2[r], EXT2), e_opn = pk_sub_i16_s(hl2[r], OPEN2); const int E = pk_max_i16(e_ext, e_opn);
                    const int f_ext = pk_sub_i16_s(f2, EXT2), f_opn = pk_sub_i16_s(hu2, OPEN2); const int F = pk_max_i16(f_ext, f_opn);
                    const int z = pk_min_u16_s(qc2[r] ^ tc2, ONE2);                                // 1 = letters differ
                    const int sc = pk_mad_i16_sv(z, NDIFF2, MATCH2) & nwq2[r] & nwt2;              // match / mismatch / 0 for wildcards
                    const int d = pk_add_i16(hd2, sc);
                    const int m1 = pk_max_i16(E, F); const int h = pk_max_i16(d, m1);
                    // COMPLEMENT flags (1 = "not equal"): bit0 h!=d, bit1 m1!=E (i.e. F>E), bit2 E!=e_ext (opened), bit3 F!=f_ext (opened).
                    // Every half holds 0/1, so plain 32-bit shifts by <= 3 stay inside their half.
                    int c = pk_min_u16_s(pk_sub_u16(h, d), ONE2);
                    c |= pk_min_u16_s(pk_sub_u16(m1, E), ONE2) << 1;
                    c |= pk_min_u16_s(pk_sub_u16(E, e_ext), ONE2) << 2;
                    c |= pk_min_u16_s(pk_sub_u16(F, f_ext), ONE2) << 3;
                    if (r < C0) acc0 = (acc0 << 4) | c; else acc1 = (acc1 << 4) | c;                // never crosses a 16-bit half: <= 4 nibbles each
                    hd2 = hl2[r];
                    hl2[r] = (h & am2) | (hl2[r] & ~am2);
                    e2[r] = E;       // not masked: before a lane's first column E only relaxes to H - open = -open (what the first real column computes from
                                     // the boundary anyway: same value, same 'opened' flag), after its last column E is not used again
                    hu2 = h; f2 = F;
                    if (r == OWN_P) cap2 = h;
                }
                // traceback word: A cells low dword, B cells high dword (complement nibbles); words of inactive steps are never read
                const unsigned wA = ((unsigned)acc0 & 0xffffu) | ((unsigned)acc1 << 16), wB = ((unsigned)acc0 >> 16) | ((unsigned)acc1 & 0xffff0000u);
                stb[(uint64_t)tau * 64 + lane] = (uint64_t)wA | ((uint64_t)wB << 32);
                const int botA_h = LO16(hu2), botA_f = LO16(f2), botB_h = HI16(hu2), botB_f = HI16(f2);
                aBot_h2 = actA ? aBot_h1 : aBot_h2; aBot_h1 = actA ? botA_h : aBot_h1; aBot_f1 = actA ? botA_f : aBot_f1; hdiagA = actA ? hupA : hdiagA;
                send_h = actB ? botB_h : send_h; send_f = actB ? botB_f : send_f;
                if (!last_strip) {
                    if (actB && lane == 63) {
                        __hip_atomic_store(&mybnd[jB], botB_h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&mybnd[bnd_stride + jB], botB_f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else {
                    const int v = own_half ? HI16(cap2) : LO16(cap2); const bool act = (own_half ? actB : actA) && lane == own_lane; const int jj = own_half ? jB : jA;
                    const bool better = act && v > bestRowV;
                    bestRowV = better ? v : bestRowV; bestRowJ = better ? jj : bestRowJ;
                }
            }
            // last target column: the state now holds H[i][m-1] for every row of the strip
#pragma unroll
            for (int r = 0; r < RP; ++r) { const int ia = i0 + r; const int v = LO16(hl2[r]); if (ia < n && v > bestColV) { bestColV = v; bestColI = ia; } }
#pragma unroll
            for (int r = 0; r < RP; ++r) { const int ib = i0 + RP + r; const int v = HI16(hl2[r]); if (ib < n && v > bestColV) { bestColV = v; bestColI = ib; } }
            if (nstrips > 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_s_waitcnt(0); }
        }
        };
        switch (own_p) {
            case 0: strips(std::integral_constant<int, 0>{}); break;
            case 1: strips(std::integral_constant<int, (RP > 1 ? 1 : 0)>{}); break;
            case 2: strips(std::integral_constant<int, (RP > 2 ? 2 : 0)>{}); break;
            case 3: strips(std::integral_constant<int, (RP > 3 ? 3 : 0)>{}); break;
            case 4: strips(std::integral_constant<int, (RP > 4 ? 4 : 0)>{}); break;
            case 5: strips(std::integral_constant<int, (RP > 5 ? 5 : 0)>{}); break;
            case 6: strips(std::integral_constant<int, (RP > 6 ? 6 : 0)>{}); break;
            default: strips(std::integral_constant<int, (RP > 7 ? 7 : 0)>{}); break;
        }
        // ---- reduce the end cell (first maximum over the last row, then strictly larger over the last column with the lowest row)
        int rowV = bestRowV, rowJ = bestRowJ;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { int ov = __shfl_xor(rowV, d), oj = __shfl_xor(rowJ, d); if (ov > rowV) { rowV = ov; rowJ = oj; } }
        int colV = bestColV, colI = bestColI;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { int ov = __shfl_xor(colV, d), oi = __shfl_xor(colI, d); if (ov > colV || (ov == colV && oi < colI)) { colV = ov; colI = oi; } }
        int ei = n - 1, ej = rowJ, best = rowV;
        if (colV > best) { best = colV; ei = colI; ej = m - 1; }

        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        // ---- traceback (uniform over the wave), identical bookkeeping to k_sg_align; only the word addressing / nibble layout differ
        // The fields of the job description that only the traceback needs are read from the kernel-argument segment HERE, through a pointer the compiler cannot see
        // through: kept in SGPRs across the step loops they pushed loop-invariant exec masks into VGPR lanes (18 v_readlane reloads per DP step, 8 % of its VALU work).
        const AlignJob* Jt = (const AlignJob*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(Jt));
        if (Jt->bp) for (int x = lane; x < Jt->bp_windows * 4; x += 64) Jt->bp[p * (uint64_t)Jt->bp_windows * 4 + x] = -1;
        {
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
            int blk_s = -1, blk_g = -1, blk_hi = -1;
            // polishing window of the current column, tracked incrementally (no divisions in the loop): [ws, ws + window), index wsn
            int wsn = bpp ? j / Jt->window : 0, ws = bpp ? wsn * Jt->window : 0;
            // (every round of the walk takes at least one step or reloads a block once per 64 steps: the bound is never reached; it turns a corrupted traceback word into a wrong
            // result the parity tests catch instead of a wave that never ends)
            for (int guard = 4 * (n + m) + 512; i >= 0 && j >= 0 && guard > 0; --guard) {
                if (bpp) while (j < ws) { ws -= Jt->window; --wsn; }
                {   // make sure the block of traceback words around the current cell is in LDS (64 steps x one group of 8 lanes)
                    const int sidx = i / STRIP; const int il = i - sidx * STRIP; const int l = il / RPL; const int rr = il - l * RPL;
                    const int tau = j + 2 * l + rr / RP; const int grp = l >> 3;
                    if (sidx != blk_s || grp != blk_g || tau > blk_hi || tau < blk_hi - 63) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        blk_s = sidx; blk_g = grp; blk_hi = tau;
                        const int tt = tau - lane;
                        if (tt >= 0) {
                            const uint4* src = (const uint4*)(mytb + ((uint64_t)sidx * steps + (uint64_t)tt) * 64 + grp * 8);
                            ngsid_v4u* dstp = (ngsid_v4u*)(tbblk + lane * 8);
                            dstp[0] = ngsid_load16_l2(src + 0); dstp[1] = ngsid_load16_l2(src + 1); dstp[2] = ngsid_load16_l2(src + 2); dstp[3] = ngsid_load16_l2(src + 3);
                        }
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                // lane k decodes the cell k diagonal steps back; in state 0 the wave takes the whole leading run of diagonal moves at once
                // (most of an alignment of similar sequences), then the first other cell is handled by the serial state machine below
                const int ik = i - lane, jk = j - lane;
                bool inb = false; int vk = 0;
                if (ik >= 0 && jk >= 0) {
                    const int sidx = ik / STRIP; const int il = ik - sidx * STRIP; const int l = il / RPL; const int rr = il - l * RPL;
                    const int half = rr / RP, r = rr - half * RP;
                    const int tau = jk + 2 * l + half;
                    if (sidx == blk_s && (l >> 3) == blk_g && tau <= blk_hi && tau >= blk_hi - 63) {
                        const uint64_t word = tbblk[(blk_hi - tau) * 8 + (l & 7)];
                        const unsigned w32 = half ? (unsigned)(word >> 32) : (unsigned)word;
                        const int sh = r < C0 ? 4 * (C0 - 1 - r) : 16 + 4 * (C1 - 1 - (r - C0));
                        vk = (int)((~(w32 >> sh)) & 15);        // stored complemented -> bit0 diag, bit1 E>=F, bit2 E extends, bit3 F extends
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
                    // window after step k: the k+1 new bits enter in step order (step 0 ends up highest)
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
        }
        __builtin_amdgcn_wave_barrier();
    }
}

#define NCLS 5
static inline uint32_t k_cls_bound_host(int c) { static const uint32_t b[NCLS] = {256, 512, 768, 896, 0xffffffffu}; return b[c]; }
__constant__ const uint32_t k_cls_bound[NCLS] = {256, 512, 768, 896, 0xffffffffu};     // query-length classes = the RP instances below

// pairs -> per-class index lists (order inside a class is irrelevant: results are written by pair index)
__global__ __launch_bounds__(256)
void k_pair_classes(AlignJob J, uint32_t* __restrict__ lists, uint32_t* __restrict__ counts, uint32_t long_len)
{
    // long_len > 0: pairs with a query or a target above it form class NCLS (the int32 kernel of k_align.hip takes them)
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int cls = -1;
    if (p < J.npairs) {
        const uint32_t qi = J.qidx[p]; const uint32_t ql = (uint32_t)(J.qoff[qi + 1] - J.qoff[qi]); cls = 0; while (ql > k_cls_bound[cls]) ++cls;
        if (long_len) { const uint32_t ti = J.tidx[p]; const uint32_t tl = (uint32_t)(J.toff[ti + 1] - J.toff[ti]); if (ql > long_len || tl > long_len) cls = NCLS; }
    }
#pragma unroll
    for (int c = 0; c < NCLS + 1; ++c) {                      // one atomic per wave and class
        const unsigned long long m = __ballot(cls == c);
        if (!m) continue;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0; if (lane == leader) base = atomicAdd(&counts[c], (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        if (cls == c) lists[(size_t)c * J.npairs + base + __popcll(m & ((1ull << lane) - 1))] = (uint32_t)p;
    }
}

struct Launch16 { uint64_t words, blocks, nwaves; uint32_t lds_per_wave, bnd_stride; int wpb; };

template <int RP>
static int32_t plan16(ngsid_ctx* ctx, uint64_t npairs, uint32_t max_qlen, uint32_t max_tlen, Launch16* L)
{
    const uint64_t strip = 128ull * RP;
    const uint64_t nstrips = (max_qlen + strip - 1) / strip;
    L->words = (nstrips ? nstrips : 1) * ((uint64_t)max_tlen + 127) * 64;
    uint64_t want = npairs;
    const uint64_t by_mem = ctx->scratch_budget / (L->words * 8 + 1);
    if (want > by_mem) want = by_mem;
    if (want < 1) want = 1;
    const uint32_t seq_lds = ((max_tlen > max_qlen ? max_tlen : max_qlen) + 15u) & ~15u;
    L->lds_per_wave = 2 * seq_lds + 4096;
    int wpb = 4;
    while (wpb > 1 && (uint64_t)wpb * L->lds_per_wave > 40 * 1024) wpb >>= 1;
    int occ = 0;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sg_align16<RP>, 64 * wpb, (size_t)wpb * L->lds_per_wave));
    if (occ < 1) occ = 1;
    const uint64_t resident = (uint64_t)occ * ctx->n_cu * wpb;          // waves that fit on the chip at once
    if (want > resident) want = resident;
    L->wpb = wpb; L->blocks = (want + wpb - 1) / wpb; L->nwaves = L->blocks * wpb;
    L->bnd_stride = (max_tlen + 15u) & ~15u;
    return NGSID_OK;
}

template <int RP>
static int32_t launch16(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, uint32_t ctr_slot = 0,
                        hipStream_t st = nullptr, uint64_t tb_off = ~0ull, uint64_t bnd_off = 0)
{
    Launch16 L; int32_t rc = plan16<RP>(ctx, job.npairs, max_qlen, max_tlen, &L); if (rc) return rc;
    if (!st) st = ctx->stream;
    if (ctx->aln_ctr.n < 16) HIPCHK(ctx, ctx->aln_ctr.alloc(16));
    if (tb_off == ~0ull) HIPCHK(ctx, hipMemsetAsync(ctx->aln_ctr.p + ctr_slot, 0, sizeof(uint32_t), st));      // (class launches: ngsid_partition_pairs has zeroed all counters)
    // scratch is grow-only and sized by the caller for all launches of a call BEFORE the first one (a reallocation frees memory
    // that an earlier, still running launch uses); concurrent class launches get their own slices (tb_off / bnd_off)
    if (tb_off == ~0ull) {
        tb_off = 0; bnd_off = 0;
        if (ctx->tb.n < L.nwaves * L.words) HIPCHK(ctx, ctx->tb.alloc(L.nwaves * L.words));
        if (ctx->bnd.n < L.nwaves * 2ull * L.bnd_stride) HIPCHK(ctx, ctx->bnd.alloc(L.nwaves * 2ull * L.bnd_stride));
    }
    { ProfScope ps_(ctx, st == ctx->stream ? "k_sg_align" : "k_sg_align_side", st);      // side-stream launches overlap the main one: timed under their own name
      hipLaunchKernelGGL((k_sg_align16<RP>), dim3((unsigned)L.blocks), dim3(64 * L.wpb), (size_t)L.wpb * L.lds_per_wave, st,
                       job, ctx->tb.p + tb_off, L.words, ctx->bnd.p + bnd_off, L.bnd_stride, L.lds_per_wave, ctx->aln_ctr.p + ctr_slot); }
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}

// the 16-bit path is exact when every score fits comfortably in int16 (see the range argument in DESIGN.md)
bool ngsid_align16_applicable(const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, int max_open)
{
    return max_qlen <= NGSID_ALIGN16_MAXLEN && max_tlen <= NGSID_ALIGN16_MAXLEN && job.match >= 0 && job.match <= 4 && job.mismatch <= 0 && job.mismatch >= -8 &&
           job.ext >= 0 && job.ext <= 4 && max_open >= 0 && max_open <= 16;
}

template <int RP>
static int32_t launch_class(ngsid_ctx* ctx, AlignJob job, int cls, uint32_t max_qlen, uint32_t max_tlen, hipStream_t st, uint64_t tb_off, uint64_t bnd_off)
{
    const uint64_t n = job.npairs;
    job.pair_list = ctx->aln_cls.p + (size_t)cls * n; job.npairs_dev = ctx->aln_ctr.p + 8 + cls;
    return launch16<RP>(ctx, job, max_qlen < k_cls_bound_host(cls) ? max_qlen : k_cls_bound_host(cls), max_tlen, (uint32_t)(1 + cls), st, tb_off, bnd_off);
}

// pairs -> NCLS index lists in ctx->aln_cls (class c at offset c * npairs), counts in ctx->aln_ctr[8 + c]; all 16 counters are zeroed first
int32_t ngsid_partition_pairs(ngsid_ctx* ctx, const AlignJob& job, uint32_t long_len)
{
    const uint64_t n = job.npairs;
    static_assert(NCLS == NGSID_ALIGN_LONG_CLASS, "the long-pair class is list NCLS");
    if (ctx->aln_ctr.n < 16) HIPCHK(ctx, ctx->aln_ctr.alloc(16));
    if (ctx->aln_cls.n < (size_t)(NCLS + 1) * n) HIPCHK(ctx, ctx->aln_cls.reserve((size_t)(NCLS + 1) * n));
    HIPCHK(ctx, hipMemsetAsync(ctx->aln_ctr.p, 0, 16 * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_pair_classes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, job, ctx->aln_cls.p, ctx->aln_ctr.p + 8, long_len);
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}

int32_t ngsid_launch_align16(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, uint32_t min_qlen, uint32_t long_len)
{
    // long_len > 0 (the caller has checked the conditions of the class path): max_qlen / max_tlen are clamped to it, longer pairs land in list NCLS for the caller
    // Large batches with mixed query lengths: split the pairs by query-length class so that every pair runs in the instance with the
    // fewest idle rows (a lane owns 2*RP rows; 750-base reads with a few 800-base ones would otherwise all run with RP = 7).
    if (job.npairs >= 4096 && max_qlen > 256 && !job.pair_list && !ngsid_opt(ctx, "align_noclass", 0)) {
        const uint64_t n = job.npairs;
        { int32_t rcp = ngsid_partition_pairs(ctx, job, long_len); if (rcp) return rcp; }
        // The class launches run CONCURRENTLY (the big class on the context's stream, the others on side streams): a class with a few hundred
        // pairs costs the latency of one pair, which would otherwise be paid once per class and call.  Every launch has its own scratch slice.
        { int32_t rs = ngsid_side_streams(ctx); if (rs) return rs; }
        uint64_t tbo[NCLS + 1] = {0}, bo[NCLS + 1] = {0};
        // the classes up to 896 bases (single strip) run two pairs per wave (k_align16p.hip) unless ngsid_ctx_option("align_paired", 0)
        const bool paired = ngsid_opt(ctx, "align_paired", 1) != 0;
        {
            Launch16 L; const uint32_t qb[NCLS] = {256, 512, 768, 896, max_qlen};
            for (int c = 0; c < NCLS; ++c) {
                const uint32_t q = std::min<uint32_t>(max_qlen, qb[c]);
                if (c == 0) plan16<2>(ctx, n, q, max_tlen, &L); else if (c == 1) plan16<4>(ctx, n, q, max_tlen, &L); else if (c == 2) plan16<6>(ctx, n, q, max_tlen, &L);
                else if (c == 3) plan16<7>(ctx, n, q, max_tlen, &L); else plan16<8>(ctx, n, q, max_tlen, &L);
                const bool used = (c == 0 || max_qlen > qb[c - 1]) && min_qlen <= qb[c];
                uint64_t words = L.nwaves * L.words, bwords = L.nwaves * 2ull * L.bnd_stride;
                if (paired && c <= 3) { int32_t rp = ngsid_paired_tb_words(ctx, c, n, max_tlen, &words); if (rp) return rp; bwords = 0; }
                tbo[c + 1] = tbo[c] + (used ? words : 0); bo[c + 1] = bo[c] + (used ? bwords : 0);
            }
            if (ctx->tb.n < tbo[NCLS]) HIPCHK(ctx, ctx->tb.reserve(tbo[NCLS]));
            if (ctx->bnd.n < bo[NCLS]) HIPCHK(ctx, ctx->bnd.reserve(bo[NCLS]));
        }
        HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        for (int i = 0; i < 4; ++i) HIPCHK(ctx, hipStreamWaitEvent(ctx->side[i], ctx->ev_fork, 0));
        int32_t rc;
        // class 2 (<= 768 bases, the ONT amplicon lengths) stays on the main stream
        if (min_qlen <= 256 && (rc = paired ? ngsid_launch_paired_class(ctx, job, 0, max_tlen, ctx->side[0], ctx->tb.p + tbo[0]) : launch_class<2>(ctx, job, 0, max_qlen, max_tlen, ctx->side[0], tbo[0], bo[0]))) return rc;
        if (max_qlen > 256 && min_qlen <= 512 && (rc = paired ? ngsid_launch_paired_class(ctx, job, 1, max_tlen, ctx->side[1], ctx->tb.p + tbo[1]) : launch_class<4>(ctx, job, 1, max_qlen, max_tlen, ctx->side[1], tbo[1], bo[1]))) return rc;
        if (max_qlen > 768 && min_qlen <= 896 && (rc = paired ? ngsid_launch_paired_class(ctx, job, 3, max_tlen, ctx->side[2], ctx->tb.p + tbo[3])
                                                                : launch_class<7>(ctx, job, 3, max_qlen, max_tlen, ctx->side[2], tbo[3], bo[3]))) return rc;
        if (max_qlen > 896 && (rc = launch_class<8>(ctx, job, 4, max_qlen, max_tlen, ctx->side[3], tbo[4], bo[4]))) return rc;
        if (max_qlen > 512 && min_qlen <= 768 && (rc = paired ? ngsid_launch_paired_class(ctx, job, 2, max_tlen, ctx->stream, ctx->tb.p + tbo[2])
                                                                : launch_class<6>(ctx, job, 2, max_qlen, max_tlen, ctx->stream, tbo[2], bo[2]))) return rc;
        for (int i = 0; i < 4; ++i) { HIPCHK(ctx, hipEventRecord(ctx->ev_join[i], ctx->side[i])); HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[i], 0)); }
        return NGSID_OK;
    }
    if (max_qlen <= 256) return launch16<2>(ctx, job, max_qlen, max_tlen);
    if (max_qlen <= 512) return launch16<4>(ctx, job, max_qlen, max_tlen);
    if (max_qlen <= 768) return launch16<6>(ctx, job, max_qlen, max_tlen);
    if (max_qlen <= 896) return launch16<7>(ctx, job, max_qlen, max_tlen);
    return launch16<8>(ctx, job, max_qlen, max_tlen);
}
