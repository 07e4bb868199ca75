// k_ed_align.hip - (a17) read -> backbone alignment of the polisher in edit-distance mode (ngsid_polish_params_t.aln_mode = 1).
//
// racon finds its window break points with edlib (unit-cost alignment, Myers' bit-vector algorithm); this is the same recurrence on
// gfx950.  Semantics = oracle ongsid_i_ed_ops (plain O(nm) DP): the whole query inside the target, target ends free, end column =
// leftmost minimum of the last row, traceback prefers diagonal, then up (query only), then left (target only).
//
// Mapping: ONE PAIR PER LANE - the recurrence has no cross-lane dependency, so a wave advances 64 alignments and every VALU
// instruction updates 64 x 64 DP cells (one 64-row block of 64 pairs).  Per target column the lane walks its query blocks top to
// bottom (Hyyro/edlib block step with a horizontal carry); the block states Pv/Mv live in registers (template BMAX blocks; longer queries
// run in groups of BMAX blocks with the horizontal deltas below a group carried through HBM), the query as three bit planes per block in LDS (letter bit 0, bit 1, "is A/C/G/T").  For the traceback every
// (block, column) stores two 64-bit vectors: DIAG (a diagonal move is optimal) and UP (vertical delta +1) - one coalesced 1 KB store
// per instruction - and each lane then walks its own path, one 16-byte L2 load per step, recording the window break points.
// ~55 VALU instructions per (block, column) for 64 pairs: ~8 k wave instructions per 750 x 750 pair instead of ~190 k in k_sg_align16.
#include "ngsid_internal.h"
#include <algorithm>

typedef unsigned long long u64;
#define LDSP __attribute__((address_space(3)))

// CLIP (round 5, ngsid_polish_params_t.aln_mode = 3): overlap-span clipping - only the columns from the first to the last run of at least CLIP_RUN equal columns are recorded
// (span and window break points), as the oracle does (ongsid_polish, clip_span): what minimap2's chain ends do before racon's edlib call.  One instance only (unbanded,
// 16-block groups): the mode serves the primer-trimming flow of the CLI, a handful of centres.
#define CLIP_RUN 15
template <int BMAX, bool WIN, bool CLIP = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 4)))
void k_ed_align(AlignJob J, ngsid_v4u* __restrict__ tb, u64 tb_per_wave /* 16-byte units */, uint32_t mstride, uint32_t* __restrict__ work_ctr, int32_t* __restrict__ dist_out,
                int8_t* __restrict__ hcar /* per wave mstride x 64: horizontal delta below the last block of a block group */,
                int bandK /* > 0: Ukkonen band for distances <= bandK (single block group only) */, uint32_t* __restrict__ fail_list, uint32_t* __restrict__ fail_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    LDSP u64* planes = (LDSP u64*)smem;                     // [block][3][lane]
    ngsid_v4u* mytb = tb + (u64)blockIdx.x * tb_per_wave;
    int8_t* myh = hcar + (u64)blockIdx.x * mstride * 64;
    const u64 npairs1 = J.npairs_dev ? (u64)*J.npairs_dev : J.npairs;      // optional indirection: a query-length class of a larger batch ...
    const u64 npairs = npairs1 + (J.npairs_dev2 ? (u64)*J.npairs_dev2 : 0);  // ... and a second class behind it
    const u64 nbundles = (npairs + 63) / 64;
    for (;;) {
        uint32_t kq = 0; if (lane == 0) kq = atomicAdd(work_ctr, 1u);
        kq = (uint32_t)__builtin_amdgcn_readfirstlane((int)kq);
        if (kq >= nbundles) break;
        const u64 pk = (u64)kq * 64 + lane;
        const bool have = pk < npairs;
        const u64 p = have ? (J.pair_list ? (pk < npairs1 ? (u64)J.pair_list[pk] : (u64)J.pair_list2[pk - npairs1]) : pk) : 0;
        const uint8_t* q = nullptr; const uint8_t* t = nullptr; int n = 0, m = 0;
        if (have) { const uint32_t qi = J.qidx[p], ti = J.tidx[p]; q = J.qseq + J.qoff[qi]; n = (int)(J.qoff[qi + 1] - J.qoff[qi]); t = J.tseq + J.toff[ti]; m = (int)(J.toff[ti + 1] - J.toff[ti]); }
        int nmax = n, mmax = m;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { nmax = max(nmax, __shfl_xor(nmax, d)); mmax = max(mmax, __shfl_xor(mmax, d)); }
        nmax = __builtin_amdgcn_readfirstlane(nmax); mmax = __builtin_amdgcn_readfirstlane(mmax);
        const int B = (nmax + 63) >> 6;                     // 64-row blocks the wave walks, in groups of BMAX (register-resident states)
        const int bl = n > 0 ? (n - 1) >> 6 : 0, lastbit = n > 0 ? (n - 1) & 63 : 0;
        int score = n, best = n, bestj = 0;                 // D[n][0] = n
        // Band (exact for pairs whose distance turns out <= bandK, the others are handed to an unbanded launch): a cell (i,c) of an alignment
        // with at most K edits has i - c <= K (its prefix costs at least i - c) and (n - i) - (m - c) <= K (the rest of the query still needs
        // that many columns), so column c only needs the rows c + min(n - m) - K ... c + K.  Blocks below the band keep their initial state
        // (vertical deltas +1, an upper bound, as in edlib); the row above the first block of the band is taken as +1 per column.
        // WIN instances (queries of any length): the register-resident blocks are a WINDOW of BMAX blocks that slides down with the band, the
        // query bit planes of a block are built when it enters the window (LDS ring), and the traceback vectors are stored band-relative
        // ([block - first block of the band in that column][column]).  A wave whose band needs more than BMAX blocks hands its pairs to the
        // unbanded launch.
        const bool band = bandK > 0 && (WIN || B <= BMAX);
        int dmin = 0, lbprev = -1, sb = 0;                  // sb: value at the bottom row of the last block of the band (per lane)
        if (band) {
            dmin = have ? n - m : 0x3fffffff;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) dmin = min(dmin, __shfl_xor(dmin, d));
            dmin = __builtin_amdgcn_readfirstlane(dmin);
        }
        if (WIN) {
            const int span = ((2 * bandK + 2 - (dmin < 0 ? dmin : 0)) >> 6) + 2;       // blocks a column's band can touch
            if (span > BMAX) {                                                             // (wave-uniform)
                if (have) fail_list[atomicAdd(fail_count, 1u)] = (uint32_t)p;
                continue;
            }
        }
        int fbw = 0;                                        // WIN: block held by register slot 0
        for (int g0 = 0; g0 < B; g0 += (WIN ? B : BMAX)) {
            const int Bg = min(BMAX, B - g0);
            // ---- query bit planes of the group (WIN: of the first BMAX blocks) -> LDS
            auto build_planes = [&](int blk, int slot) {
                u64 lo = 0, hi = 0, ok = 0;
                const int base = blk * 64;
                for (int r = 0; r < 64; ++r) {
                    const int i = base + r;
                    const int c = i < n ? ngsid_bcode(q[i]) : 4;
                    lo |= (u64)(c & 1) << r; hi |= (u64)((c >> 1) & 1) << r; ok |= (u64)(c < 4) << r;
                }
                planes[(slot * 3 + 0) * 64 + lane] = lo; planes[(slot * 3 + 1) * 64 + lane] = hi; planes[(slot * 3 + 2) * 64 + lane] = ok;
            };
            for (int b = 0; b < Bg; ++b) build_planes(g0 + b, b);
            u64 Pv[BMAX], Mv[BMAX];
#pragma unroll
            for (int b = 0; b < BMAX; ++b) { Pv[b] = ~0ull; Mv[b] = 0ull; }
            const bool more = !WIN && g0 + BMAX < B;        // a further group follows: keep the horizontal deltas of this group's last row
            // ---- forward: column by column, blocks top to bottom.  The target letters of 64 columns are three bit planes in REGISTERS (letter bit 0, bit 1, "is A/C/G/T"),
            // refilled every 64 columns: the column loop itself contains no global load, so no column waits (s_waitcnt vmcnt) for the traceback-word stores of
            // the columns before it.  The horizontal delta handed from block to block travels as two 0/1 words (hp: +1, hm: -1): no per-lane branches in a block step.
            u64 TPlo = 0, TPhi = 0, TPok = 0;
            for (int j = 0; j < mmax; ++j) {
                const int jb = j & 63;
                if (jb == 0) {
                    u64 lo = 0, hi = 0, ok = 0;
#pragma unroll 16
                    for (int c = 0; c < 64; ++c) {
                        const int jc = j + c;
                        const int cd = jc < m ? ngsid_bcode(t[jc]) : 4;
                        lo |= (u64)(cd & 1) << c; hi |= (u64)((cd >> 1) & 1) << c; ok |= (u64)(cd < 4) << c;
                    }
                    TPlo = lo; TPhi = hi; TPok = ok;
                }
                const u64 Tlo = 0ull - ((TPlo >> jb) & 1ull), Thi = 0ull - ((TPhi >> jb) & 1ull), Tok = 0ull - ((TPok >> jb) & 1ull);
                int fb = 0, lb = Bg - 1;
                if (band) {
                    const int lo_row = j + dmin - bandK - 1;
                    fb = lo_row > 0 ? (lo_row >> 6) : 0; lb = min(B - 1, (j + 1 + bandK) >> 6);
                    if (lb > lbprev) {          // blocks entering the band: the last row of a lane may be among them
                        if (n > 0 && bl > lbprev && bl <= lb) score = sb + 64 * (bl - 1 - lbprev) + lastbit + 1;
                        sb += 64 * (lb - lbprev);
                    }
                }
                if (WIN) {
                    while (fb > fbw) {          // the band has left block fbw: slide the register window, build the planes of the block that enters
#pragma unroll
                        for (int b = 0; b + 1 < BMAX; ++b) { Pv[b] = Pv[b + 1]; Mv[b] = Mv[b + 1]; }
                        Pv[BMAX - 1] = ~0ull; Mv[BMAX - 1] = 0ull;
                        ++fbw;
                        const int nb = fbw + BMAX - 1;
                        if (nb < B) build_planes(nb, nb % BMAX);
                    }
                }
                unsigned hp, hm;                                                     // horizontal delta entering the next block: +1 / -1 / 0 as (hp, hm) = (1,0) / (0,1) / (0,0)
                if (g0) { const int h0 = (int)myh[(u64)j * 64 + lane]; hp = h0 > 0; hm = h0 < 0; }
                else { hp = fb > 0 ? 1u : 0u; hm = 0u; }                             // top row of the matrix is all zeros (target prefix free)
                ngsid_v4u* col = mytb + ((u64)j * 64 + lane);
                const int slot0 = WIN ? fbw % BMAX : 0;
                const bool inm = j < m;
#pragma unroll
                for (int b = 0; b < BMAX; ++b) {
                    const int blk = WIN ? fbw + b : g0 + b;                          // block held by register slot b
                    if (WIN ? blk <= lb : (b >= fb && b <= lb)) {                    // WIN: slot 0 holds the first block of the band (fbw == fb)
                        const int ps = WIN ? (slot0 + b >= BMAX ? slot0 + b - BMAX : slot0 + b) : b;
                        const u64 lo = planes[(ps * 3 + 0) * 64 + lane], hi = planes[(ps * 3 + 1) * 64 + lane], ok = planes[(ps * 3 + 2) * 64 + lane];
                        const u64 Eq = ~((lo ^ Tlo) | (hi ^ Thi)) & ok & Tok;
                        const u64 pv = Pv[b], mv = Mv[b];
                        const u64 Xv = Eq | mv;
                        const u64 Eqh = Eq | (u64)hm;
                        const u64 Xh = (((Eqh & pv) + pv) ^ pv) | Eqh;
                        u64 Ph = mv | ~(Xh | pv), Mh = pv & Xh;
                        // moves (oracle: diagonal if D[i-1][j-1] + neq == D[i][j]): a match always qualifies; a mismatch iff the diagonal delta
                        // h(i,j) + v(i,j-1) is +1, i.e. (h,v) = (+1,0) or (0,+1)
                        const u64 diag = Eq | (Ph & ~(pv | mv)) | (~(Ph | Mh) & pv);
                        if (blk == bl && inm) {
                            score += (int)((Ph >> lastbit) & 1) - (int)((Mh >> lastbit) & 1);
                            if (score < best) { best = score; bestj = j + 1; }
                        }
                        const unsigned php = (unsigned)(Ph >> 63), phm = (unsigned)(Mh >> 63);
                        Ph = (Ph << 1) | (u64)hp; Mh = (Mh << 1) | (u64)hm;
                        const u64 npv = Mh | ~(Xv | Ph);
                        Pv[b] = npv; Mv[b] = Ph & Xv;
                        ngsid_v4u w; w.x = (unsigned)diag; w.y = (unsigned)(diag >> 32); w.z = (unsigned)npv; w.w = (unsigned)(npv >> 32);
#if defined(ED_EXP_STORE_EVERY)
                        if ((j % ED_EXP_STORE_EVERY) == 0)           // timing build (wrong results): what the traceback-word stores cost
#endif
                        col[(u64)(WIN ? b : blk) * mstride * 64] = w;
                        hp = php; hm = phm;
                    }
                }
                if (more) myh[(u64)j * 64 + lane] = (int8_t)((int)hp - (int)hm);
                if (band) { sb += (int)hp - (int)hm; lbprev = lb; }
            }
            if (more) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // ---- traceback, every lane on its own path
        bool ok = true;
        if (band && have && best > bandK) { ok = false; fail_list[atomicAdd(fail_count, 1u)] = (uint32_t)p; }
        int i = ok ? n : 0, j = bestj;
        int q_end = -1, t_end = -1, q_beg = -1, t_beg = -1;
        int32_t* bpp = (J.bp && have) ? J.bp + p * (u64)J.bp_windows * 4 : nullptr;
        if (bpp) for (int x = 0; x < J.bp_windows * 4; ++x) bpp[x] = -1;
        const int W = J.window > 0 ? J.window : 0x7fffffff;
        int cw = -1, w_qf = 0, w_ql = 0, w_tf = 0, w_tl = 0;
        int c_run = 0, c_x0q = -1, c_x0t = -1; bool c_started = false;       // CLIP state (unused otherwise)
        int wsn = j > 0 ? (j - 1) / W : 0, ws = wsn * W;       // window of the current target position, tracked without divisions
        // The lanes walk their paths IN LOCKSTEP OVER THE TARGET COLUMNS (round 3): in round jj every lane whose path stands in column jj handles that column (any
        // number of vertical moves, then one diagonal or horizontal move), so the 64 loads of a round go to the same column - one or two coalesced KB instead of
        // 64 scattered sectors (the paths of a bundle drift apart by a few columns, which made every lane fetch its own sector) - and the word of the next column
        // is requested one round ahead, for the block the path most likely reaches.  A path that is at j = 0 only has vertical moves left: nothing to record.
        auto tb_blk = [&](int ii, int col0) { int tb_ = (ii - 1) >> 6; if (WIN) { const int lo_row = col0 + dmin - bandK - 1; tb_ -= lo_row > 0 ? (lo_row >> 6) : 0; } return tb_; };
        auto tb_load = [&](int tb_, int col0) { return __builtin_nontemporal_load(mytb + (((u64)tb_ * mstride + (u64)col0) * 64 + lane)); };
        int jj = i > 0 ? j : 0;
#if defined(ED_EXP_NO_TB)
        jj = 0;                                                       // timing experiment (wrong results): no traceback
#endif
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) jj = max(jj, __shfl_xor(jj, d));
        jj = __builtin_amdgcn_readfirstlane(jj);
        ngsid_v4u wn; wn.x = wn.y = wn.z = wn.w = 0; int wn_blk = -0x7fffffff;
        if (i > 0 && j == jj && jj >= 1) { wn_blk = tb_blk(i, jj - 1); wn = tb_load(wn_blk, jj - 1); }
        for (; jj >= 1; --jj) {
            if (!__ballot(i > 0 && j > 0)) break;
            const bool on = i > 0 && j == jj;
            ngsid_v4u w = wn; int wb = wn_blk;
            wn_blk = -0x7fffffff;
            if (jj >= 2) {                  // request the word of column jj - 2 for the next round
                int gi = 0;
                if (on && i > 1) gi = i - 1;                         // most moves are diagonal
                else if (i > 0 && j == jj - 1) gi = i;               // a path that starts in the next round
                if (gi > 0) { wn_blk = tb_blk(gi, jj - 2); wn = tb_load(wn_blk, jj - 2); }
            }
            if (on) {
                { const int nb = tb_blk(i, jj - 1); if (nb != wb) { w = tb_load(nb, jj - 1); wb = nb; } }
                for (;;) {
                    const int bit = (i - 1) & 63;
                    const u64 dv = ((u64)w.y << 32) | w.x, uv = ((u64)w.w << 32) | w.z;
                    const bool diag = (dv >> bit) & 1, up = (uv >> bit) & 1;
                    if (diag) {
                        const int qi0 = i - 1, ti0 = j - 1;
                        auto record = [&](int qi, int ti) {
                            if (q_end < 0) { q_end = qi; t_end = ti; }
                            q_beg = qi; t_beg = ti;
                            if (bpp) {
                                while (ti < ws) { ws -= W; --wsn; }
                                const int wn_ = wsn;
                                if (wn_ != cw) { if (cw >= 0 && cw < J.bp_windows) { bpp[cw * 4 + 0] = w_qf; bpp[cw * 4 + 1] = w_ql; bpp[cw * 4 + 2] = w_tf; bpp[cw * 4 + 3] = w_tl; } cw = wn_; w_ql = qi; w_tl = ti; }
                                w_qf = qi; w_tf = ti;
                            }
                        };
                        if constexpr (CLIP) {
                            const uint8_t cq = q[qi0] & 0xDF, ct = t[ti0] & 0xDF;              // equal letters of A/C/G/T, any case (oracle ed_code)
                            const bool eq = cq == ct && (cq == 'A' || cq == 'C' || cq == 'G' || cq == 'T');
                            c_run = eq ? c_run + 1 : 0;
                            if (!c_started) {
                                if (c_run >= CLIP_RUN) {        // the LAST run of the alignment (walking backwards, the first): its end lies CLIP_RUN - 1 columns behind - record the run
                                    c_started = true;
                                    for (int d = CLIP_RUN - 1; d >= 0; --d) record(qi0 + d, ti0 + d);
                                    c_x0q = qi0; c_x0t = ti0;
                                }
                            } else {
                                record(qi0, ti0);
                                if (c_run >= CLIP_RUN) { c_x0q = qi0; c_x0t = ti0; }        // inside a long run: the earliest such column so far
                            }
                        } else record(qi0, ti0);
                        --i; --j; break;
                    }
                    if constexpr (CLIP) c_run = 0;                                              // a gap column ends a run
                    if (!up) { --j; break; }
                    --i;
                    if (i == 0) break;
                    { const int nb = tb_blk(i, jj - 1); if (nb != wb) { w = tb_load(nb, jj - 1); wb = nb; } }
                }
            }
        }
        i = 0;
        if (bpp && ok && cw >= 0 && cw < J.bp_windows) { bpp[cw * 4 + 0] = w_qf; bpp[cw * 4 + 1] = w_ql; bpp[cw * 4 + 2] = w_tf; bpp[cw * 4 + 3] = w_tl; }
        if constexpr (CLIP) {
            if (have && ok && c_started) {      // head side: everything in front of the first run of CLIP_RUN equal columns goes
                q_beg = c_x0q; t_beg = c_x0t;
                if (bpp) { const int w0 = c_x0t / W; for (int x = 0; x < w0 && x < J.bp_windows; ++x) { bpp[x * 4 + 0] = -1; bpp[x * 4 + 1] = -1; bpp[x * 4 + 2] = -1; bpp[x * 4 + 3] = -1; }
                           if (w0 < J.bp_windows) { bpp[w0 * 4 + 0] = c_x0q; bpp[w0 * 4 + 2] = c_x0t; } }
            }
        }
        if (have && ok) {
            if (dist_out) dist_out[p] = best;
            if (J.span) { J.span[p * 4 + 0] = q_beg; J.span[p * 4 + 1] = q_end; J.span[p * 4 + 2] = t_beg; J.span[p * 4 + 3] = t_end; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Traceback scratch of `want` resident waves; when the device is short of memory (other contexts / processes on the same GPU) the number of
// resident waves is halved until the block fits - the persistent waves pull pairs from a queue, so fewer of them only lower the parallelism.
static int32_t ed_reserve(ngsid_ctx* ctx, u64& want, u64 per_wave)
{
    for (;;) {
        if (ctx->ed_tb.n >= want * per_wave) return NGSID_OK;
        hipError_t e = ctx->ed_tb.reserve(want * per_wave);
        if (e == hipSuccess) return NGSID_OK;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || want <= 64) HIPCHK(ctx, e);
        want = std::max<u64>(64, want / 2);
    }
}

template <int BMAX, bool WIN = false, bool CLIP = false>
static int32_t launch_ed(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, int32_t* dist_out, uint32_t ctr_slot = 14, int bandK = 0, uint32_t fail_slot = 15, uint32_t* fail_list = nullptr)
{
    const u64 nbundles = (job.npairs + 63) / 64;
    const uint32_t mstride = (max_tlen + 63u) & ~63u;                     // rounded so that backbones growing by a few bases between iterations reuse the scratch
    const u64 nblocks = WIN ? (u64)BMAX : std::max<u64>(1, ((u64)max_qlen + 63) / 64);     // WIN: band-relative storage, BMAX blocks per column
    const u64 per_wave = nblocks * mstride * 64;                           // 16-byte units
    const size_t lds = (size_t)BMAX * 3 * 64 * 8 + (size_t)ngsid_opt(ctx, "ed_lds_pad_kb", 0) * 1024;      // (dev option: extra LDS per wave = fewer resident waves, for occupancy measurements)
    int occ = 0;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_ed_align<BMAX, WIN, CLIP>, 64, lds));
    if (occ < 1) occ = 1;
    u64 want = std::min<u64>(nbundles, (u64)occ * ctx->n_cu);
    const u64 by_mem = std::max<u64>(1, std::min<size_t>((size_t)24 << 30, ctx->scratch_budget) / (per_wave * 16));
    want = std::max<u64>(1, std::min(want, by_mem));
    { int32_t rr = ed_reserve(ctx, want, per_wave); if (rr) return rr; }
    if (ctx->ed_h.n < want * (u64)mstride * 64) HIPCHK(ctx, ctx->ed_h.reserve(want * (u64)mstride * 64));
    if (ctx->aln_ctr.n < 16) HIPCHK(ctx, ctx->aln_ctr.alloc(16));
    HIPCHK(ctx, hipMemsetAsync(ctx->aln_ctr.p + ctr_slot, 0, sizeof(uint32_t), ctx->stream));
    { ProfScope ps_(ctx, "k_ed_align"); hipLaunchKernelGGL((k_ed_align<BMAX, WIN, CLIP>), dim3((unsigned)want), dim3(64), lds, ctx->stream, job, ctx->ed_tb.p, per_wave, mstride, ctx->aln_ctr.p + ctr_slot, dist_out, ctx->ed_h.p,
                                                            bandK, fail_list ? fail_list : ctx->ed_fail.p, ctx->aln_ctr.p + fail_slot); }
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}

// pairs whose distance exceeded the band of the first launch (list ctx->ed_fail, count aln_ctr[15]): unbanded.  With retryK > 0 (class launches with a band of at most
// 150) they first run in the sliding-window instance with that wider band - half the blocks per column of the unbanded instance, and a handful of pairs costs the
// latency of ONE pair - and only what exceeds it too (list ctx->ed_fail2, count aln_ctr[7]) runs unbanded.
static int32_t launch_ed_fallback(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, int32_t* dist_out, int retryK = 0)
{
    AlignJob j = job; j.pair_list = ctx->ed_fail.p; j.npairs_dev = ctx->aln_ctr.p + 15; j.pair_list2 = nullptr; j.npairs_dev2 = nullptr;
    if (retryK > 0) {
        HIPCHK(ctx, ctx->ed_fail2.reserve(job.npairs));
        int32_t rc = launch_ed<8, true>(ctx, j, max_qlen, max_tlen, dist_out, 6, retryK, 7, ctx->ed_fail2.p);
        if (rc) return rc;
        j.pair_list = ctx->ed_fail2.p; j.npairs_dev = ctx->aln_ctr.p + 7;
    }
    return launch_ed<16>(ctx, j, max_qlen, max_tlen, dist_out, 13, 0);
}

int32_t ngsid_launch_ed_align(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, int32_t* dist_out)
{
    if (job.npairs == 0) return NGSID_OK;
    if (max_qlen > NGSID_MAX_CONSENSUS_LEN || max_tlen > NGSID_MAX_CONSENSUS_LEN) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "sequence longer than %d in the edit-distance aligner", NGSID_MAX_CONSENSUS_LEN);
    if (job.clip) return launch_ed<16, false, true>(ctx, job, max_qlen, max_tlen, dist_out, 14, 0);      // overlap-span clipping (aln_mode 3): the one CLIP instance, unbanded
    // band: wide enough for the usual read-to-draft distance, pairs beyond it take the unbanded launch (the result does not depend on it)
    int bandK = 64 + (int)(max_qlen / 32);
    const bool band_set = ngsid_opt(ctx, "ed_band", -1) >= 0;
    if (band_set) bandK = (int)ngsid_opt(ctx, "ed_band", -1);
    const bool win = max_qlen > 1024 && bandK > 0;  // long queries: sliding window of 8 register-resident blocks (band of at most ~380 rows) ...
    bool win16 = false;                             // ... or of 16 (~900 rows) when the reads are so long that a 5 % error rate needs it
    if (win) {
        if (bandK > 150 && !band_set) { win16 = true; bandK = std::min(64 + (int)(max_qlen / 16), 400); }
        else if (bandK > 150) { win16 = true; bandK = std::min(bandK, 400); }
    }
    if (ctx->aln_ctr.n < 16) HIPCHK(ctx, ctx->aln_ctr.alloc(16));
    if (bandK > 0) HIPCHK(ctx, ctx->ed_fail.reserve(job.npairs));
    if (job.npairs >= 4096 && max_qlen > 256 && !job.pair_list && !ngsid_opt(ctx, "align_noclass", 0)) {
        // mixed lengths: every pair runs in the instance with the fewest register-resident blocks that holds its query
        // (the scratch is sized by the longest query; launches of one call share it, the stream serialises them)
        int32_t rc = ngsid_partition_pairs(ctx, job); if (rc) return rc;
        const u64 n = job.npairs;
        auto cls = [&](int c, uint32_t bound) { AlignJob j = job; j.pair_list = ctx->aln_cls.p + (size_t)c * n; j.npairs_dev = ctx->aln_ctr.p + 8 + c; (void)bound; return j; };
        {   // reserve the scratch once, for the class with the largest footprint (16-block instance, longest query)
            const uint32_t mstride = (max_tlen + 63u) & ~63u; const u64 nblocks = std::max<u64>(1, ((u64)max_qlen + 63) / 64); const u64 per_wave = nblocks * mstride * 64;
            const u64 want = std::max<u64>(1, std::min<u64>((u64)8 * ctx->n_cu, std::max<u64>(1, std::min<size_t>((size_t)24 << 30, ctx->scratch_budget) / (per_wave * 16))));
            u64 w2 = want; { int32_t rr = ed_reserve(ctx, w2, per_wave); if (rr) return rr; }
            if (ctx->ed_h.n < w2 * (u64)mstride * 64) HIPCHK(ctx, ctx->ed_h.reserve(w2 * (u64)mstride * 64));
        }
        if ((rc = launch_ed<4>(ctx, cls(0, 256), std::min<uint32_t>(max_qlen, 256), max_tlen, dist_out, 1, bandK))) return rc;
        if (max_qlen > 256 && (rc = launch_ed<8>(ctx, cls(1, 512), std::min<uint32_t>(max_qlen, 512), max_tlen, dist_out, 2, bandK))) return rc;
        // 513-768 bases: the 8-block window instance as well (12 KB of LDS per wave instead of 18 KB: three waves per SIMD instead of two, -8 %);
        // NGSID_ED_WIN_ALL=0 selects the 12-block instance with all blocks resident
        const bool win_all = ngsid_opt(ctx, "ed_win_all", 1) != 0;
        // The sliding-window instance takes queries of any length, so the 769-896 class (the few reads of a 750-base amplicon set that came out long) rides in the same
        // launch as a second index list instead of paying the latency of one pair in a launch of its own (2 ms per polishing iteration at C3).
        const bool win2 = win_all && bandK > 0 && bandK <= 150;
        if (max_qlen > 512) {
            AlignJob j2 = cls(2, 768);
            if (win2 && max_qlen > 768) { j2.pair_list2 = ctx->aln_cls.p + (size_t)3 * n; j2.npairs_dev2 = ctx->aln_ctr.p + 8 + 3; }
            // a band of up to ~100 fits a window of SIX blocks (9 KB of LDS and 12 state registers less per wave: four waves per SIMD instead of three); a bundle
            // whose length spread needs more (span check in the kernel) goes to the retry launch
            const bool win6 = win2 && bandK <= 100 && ngsid_opt(ctx, "ed_win6", 1) != 0;
            if ((rc = win6 ? launch_ed<6, true>(ctx, j2, std::min<uint32_t>(max_qlen, 896), max_tlen, dist_out, 3, bandK)
                    : win2 ? launch_ed<8, true>(ctx, j2, std::min<uint32_t>(max_qlen, 896), max_tlen, dist_out, 3, bandK)
                           : launch_ed<12>(ctx, j2, std::min<uint32_t>(max_qlen, 768), max_tlen, dist_out, 3, bandK))) return rc;
        }
        if (max_qlen > 768 && !win2 && (rc = launch_ed<16>(ctx, cls(3, 896), std::min<uint32_t>(max_qlen, 896), max_tlen, dist_out, 4, bandK))) return rc;
        if (max_qlen > 896 && (rc = win16 ? launch_ed<16, true>(ctx, cls(4, 0), max_qlen, max_tlen, dist_out, 5, bandK) : win ? launch_ed<8, true>(ctx, cls(4, 0), max_qlen, max_tlen, dist_out, 5, bandK) : launch_ed<16>(ctx, cls(4, 0), max_qlen, max_tlen, dist_out, 5, bandK))) return rc;
        return bandK > 0 ? launch_ed_fallback(ctx, job, max_qlen, max_tlen, dist_out, win2 ? 180 : 0) : NGSID_OK;
    }
    if (bandK > 0) HIPCHK(ctx, hipMemsetAsync(ctx->aln_ctr.p + 15, 0, sizeof(uint32_t), ctx->stream));
    int32_t rc;
    if (max_qlen <= 256) rc = launch_ed<4>(ctx, job, max_qlen, max_tlen, dist_out, 14, bandK);
    else if (max_qlen <= 512) rc = launch_ed<8>(ctx, job, max_qlen, max_tlen, dist_out, 14, bandK);
    else if (max_qlen <= 768) rc = launch_ed<12>(ctx, job, max_qlen, max_tlen, dist_out, 14, bandK);
    else if (win16) rc = launch_ed<16, true>(ctx, job, max_qlen, max_tlen, dist_out, 14, bandK);
    else if (win) rc = launch_ed<8, true>(ctx, job, max_qlen, max_tlen, dist_out, 14, bandK);
    else rc = launch_ed<16>(ctx, job, max_qlen, max_tlen, dist_out, 14, bandK);          // longer queries: groups of 16 blocks, horizontal deltas carried through HBM
    if (rc || bandK <= 0) return rc;
    return launch_ed_fallback(ctx, job, max_qlen, max_tlen, dist_out);
}
