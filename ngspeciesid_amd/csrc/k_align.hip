// k_align.hip - (a10,a15) batched semi-global affine alignment WITH traceback, one 64-lane wave per pair.
//
// Replaces parasail.sg_trace_scan_16/32 + cigar_to_seq + the k-column identity windows of
// cluster.py:130-169, and the identity count of consensus.py:129-145.  Also produces racon-style window
// break points for the polisher (replaces the edlib NW walk of racon's overlap.cpp).
//
// Mapping to CDNA4: lane l owns RPL consecutive query rows; the wave sweeps target columns as a systolic
// anti-diagonal (lane l works on column tau-l at step tau), the row-boundary (H,F) pair moves to the next
// lane with one cross-lane shift per step, H/E of the lane's rows live in VGPRs, the target is staged in
// LDS.  Every step each lane emits 4 traceback bits per cell (2 H-source, 1 E-extend, 1 F-extend) packed
// in one 64-bit word -> a fully coalesced 512 B store per wave per step.  The traceback then walks the
// words backwards and folds the k-column window statistic on the fly in a 64-bit shift register
// (the window count is symmetric under path reversal), so no CIGAR or gapped strings ever exist in HBM.
// Integer work throughout (int32 scores); bound by VALU issue, not by HBM: ~0.5 KB of traceback per
// DP step versus ~300 VALU ops.
#include "ngsid_internal.h"

#define NEGINF (-(1 << 29))

template <int RPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4)))
void k_sg_align(AlignJob J, uint64_t* __restrict__ tb, uint64_t tb_words_per_wave, int32_t* __restrict__ bnd, uint32_t bnd_stride, uint32_t lds_per_wave, uint32_t* __restrict__ work_ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    uint8_t* tgt = smem + (size_t)wib * lds_per_wave;                 // raw target, raw query, 4 KB traceback block
    const uint32_t seq_lds = (lds_per_wave - 4096) / 2;
    uint8_t* qry = tgt + seq_lds;
    uint64_t* tbblk = (uint64_t*)(tgt + 2 * (size_t)seq_lds);
    uint64_t* mytb = tb + wave * tb_words_per_wave;
    int32_t* mybnd = bnd + wave * (uint64_t)bnd_stride * 2;
    const int STRIP = 64 * RPL;

    for (;;) {
        // persistent waves pull pairs from a queue: the grid is sized to what is resident, so there is no tail of idle SIMDs
        uint32_t pq = 0; if (lane == 0) pq = atomicAdd(work_ctr, 1u);
        const uint64_t px = (uint32_t)__builtin_amdgcn_readfirstlane((int)pq);
        if (px >= (J.npairs_dev ? (uint64_t)*J.npairs_dev : J.npairs)) break;
        const uint64_t p = J.pair_list ? (uint64_t)J.pair_list[px] : px;       // (round 5: the long-pair class of a partitioned batch comes as an index list with its count on the device)
        const uint32_t qi = J.qidx[p], ti = J.tidx[p];
        const uint8_t* q = J.qseq + J.qoff[qi]; const int n = (int)(J.qoff[qi + 1] - J.qoff[qi]);
        const uint8_t* t = J.tseq + J.toff[ti]; const int m = (int)(J.toff[ti + 1] - J.toff[ti]);
        const int gopen = J.open[p], gext = J.ext, smatch = J.match, smis = J.mismatch;
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
        for (int x = lane; x < m; x += 64) tgt[x] = t[x];
        for (int x = lane; x < n; x += 64) qry[x] = q[x];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();

        const int steps = m + 63;
        const int nstrips = (n + STRIP - 1) / STRIP;
        int bestRowV = NEGINF, bestRowJ = 0;       // last query row (ascending column, first maximum)
        int bestColV = NEGINF, bestColI = 0x7fffffff;   // last target column (ascending row, strictly larger only)

        for (int sidx = 0; sidx < nstrips; ++sidx) {
            const int i0 = sidx * STRIP + lane * RPL;
            int qc[RPL], hl[RPL], e[RPL];
#pragma unroll
            for (int r = 0; r < RPL; ++r) { qc[r] = (i0 + r < n) ? ngsid_bcode(qry[i0 + r]) : 4; hl[r] = 0; e[r] = NEGINF; }
            const int rlast = (n - 1) - i0;            // row n-1 lives in this lane iff 0 <= rlast < RPL
            int hdiag_top = 0;                         // H[i0-1][j-1]
            int send_h = 0, send_f = NEGINF;
            uint64_t* stb = mytb + (uint64_t)sidx * steps * 64;
            for (int tau = 0; tau < steps; ++tau) {
                const int j = tau - lane;
                // lane l-1 -> lane l in one DPP move (wave_shr:1), no LDS crossbar round trip
                int hup = __builtin_amdgcn_update_dpp(0, send_h, 0x138, 0xf, 0xf, false), fup = __builtin_amdgcn_update_dpp(0, send_f, 0x138, 0xf, 0xf, false);
                if (lane == 0) {
                    if (sidx == 0) { hup = 0; fup = NEGINF; }
                    else if (j >= 0 && j < m) {
                        hup = __hip_atomic_load(&mybnd[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        fup = __hip_atomic_load(&mybnd[bnd_stride + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (j >= 0 && j < m) {
                    const int tc = ngsid_bcode(tgt[j]);
                    int hd = hdiag_top, hu = hup, f = fup;
                    uint64_t word = 0;
#pragma unroll
                    for (int r = 0; r < RPL; ++r) {
                        const int e_ext = e[r] - gext, e_opn = hl[r] - gopen;
                        const int ebit = e_ext >= e_opn; const int E = ebit ? e_ext : e_opn;
                        const int f_ext = f - gext, f_opn = hu - gopen;
                        const int fbit = f_ext >= f_opn; const int F = fbit ? f_ext : f_opn;
                        const int a = qc[r];
                        const int sc = ((a | tc) > 3) ? 0 : (a == tc ? smatch : smis);
                        const int d = hd + sc;
                        int h, src;
                        if (d >= E && d >= F) { h = d; src = 0; } else if (E >= F) { h = E; src = 1; } else { h = F; src = 2; }
                        word |= (uint64_t)(src | (ebit << 2) | (fbit << 3)) << (4 * r);
                        hd = hl[r]; hl[r] = h; e[r] = E; hu = h; f = F;
                    }
                    hdiag_top = hup;
                    send_h = hu; send_f = f;
                    stb[(uint64_t)tau * 64 + lane] = word;
                    // lane 63 hands the strip's bottom row to the next strip through HBM.  In-place is safe: lane 0 of
                    // this strip consumed bnd[j] 63 steps ago.  Agent-scope relaxed accesses keep the hand-off out of L1.
                    if (lane == 63 && sidx + 1 < nstrips) {
                        __hip_atomic_store(&mybnd[j], hu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(&mybnd[bnd_stride + j], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (rlast >= 0 && rlast < RPL) {
                        int v = hl[0];
#pragma unroll
                        for (int r = 1; r < RPL; ++r) if (r == rlast) v = hl[r];
                        if (v > bestRowV) { bestRowV = v; bestRowJ = j; }
                    }
                    if (j == m - 1) {
#pragma unroll
                        for (int r = 0; r < RPL; ++r) if (i0 + r < n && hl[r] > bestColV) { bestColV = hl[r]; bestColI = i0 + r; }
                    }
                }
            }
            if (nstrips > 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __builtin_amdgcn_s_waitcnt(0); }
        }
        // ---- reduce the end cell
        // last row: exactly one lane (owner of row n-1 in the last strip) holds a value
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

        // ---- traceback: executed uniformly by the whole wave; traceback words are pulled 64 steps x 8 lanes (4 KB) at a time
        //      into LDS (one HBM round trip per ~64 path steps instead of one per step), sequences are read from LDS.
        if (J.bp) for (int x = lane; x < J.bp_windows * 4; x += 64) J.bp[p * (uint64_t)J.bp_windows * 4 + x] = -1;
        {
            const int K = J.k; const int mid = J.match_id ? J.match_id[p] : K;
            const uint64_t kmask = (K >= 64) ? ~0ull : ((1ull << K) - 1);
            uint64_t win = 0; int cols = 0, nm = 0, region = 0;
            uint8_t* opp = J.ops ? J.ops + J.ops_off[p] : nullptr; int oc = 0;      // optional: the alignment columns themselves, in traceback (reverse) order
            if (opp && lane == 0) { for (int x = n - 1; x > ei; --x) opp[oc++] = 2; for (int x = m - 1; x > ej; --x) opp[oc++] = 3; }
            {   // trailing end gaps (walked first)
                const int z = (n - 1 - ei) + (m - 1 - ej);
                const int zl = z < K ? z : K;             // after K zeros the window is all zero
                for (int x = 0; x < zl; ++x) { win <<= 1; ++cols; if (cols >= K) region += ((int)__popcll(win & kmask) >= mid); }
                if (z > zl) { region += (0 >= mid) ? (z - zl) : 0; cols += z - zl; }
            }
            int i = ei, j = ej, state = 0;
            int q_end = -1, t_end = -1, q_beg = -1, t_beg = -1;
            int cw = -1, w_qf = 0, w_ql = 0, w_tf = 0, w_tl = 0;
            int32_t* bpp = J.bp ? J.bp + p * (uint64_t)J.bp_windows * 4 : nullptr;
            int blk_s = -1, blk_g = -1, blk_hi = -1;       // loaded block: strip, 8-lane group, highest step
            // (every round of the walk takes at least one step or reloads a block once per 64 steps: the bound is never reached; it turns a corrupted traceback word into a wrong
            // result the parity tests catch instead of a wave that never ends)
            for (int guard = 4 * (n + m) + 512; i >= 0 && j >= 0 && guard > 0; --guard) {
                const int sidx = i / STRIP; const int il = i - sidx * STRIP; const int l = il / RPL; const int r = il - l * RPL;
                const int tau = j + l; const int grp = l >> 3;
                if (sidx != blk_s || grp != blk_g || tau > blk_hi || tau < blk_hi - 63) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    blk_s = sidx; blk_g = grp; blk_hi = tau;
                    const int tt = tau - lane;
                    if (tt >= 0) {
                        const uint4* src = (const uint4*)(mytb + ((uint64_t)sidx * steps + (uint64_t)tt) * 64 + grp * 8);
                        ngsid_v4u* dstp = (ngsid_v4u*)(tbblk + lane * 8);
                        // nt loads are served by L2: this wave rewrites the same scratch addresses for every pair, an L1 line may be stale
                        dstp[0] = ngsid_load16_l2(src + 0); dstp[1] = ngsid_load16_l2(src + 1); dstp[2] = ngsid_load16_l2(src + 2); dstp[3] = ngsid_load16_l2(src + 3);
                    }
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
                const uint64_t word = tbblk[(blk_hi - tau) * 8 + (l & 7)];
                const int v = (int)((word >> (4 * r)) & 15);
                int bit = 0, emit = 1;
                if (state == 0) {
                    const int src = v & 3;
                    if (src == 0) {
                        bit = (qry[i] == tgt[j]);
                        if (opp && lane == 0) opp[oc++] = bit ? 0 : 1;
                        if (q_end < 0) { q_end = i; t_end = j; }
                        q_beg = i; t_beg = j;
                        if (bpp) {
                            const int wn = j / J.window;
                            if (wn != cw) { if (lane == 0 && cw >= 0 && cw < J.bp_windows) { bpp[cw * 4 + 0] = w_qf; bpp[cw * 4 + 1] = w_ql; bpp[cw * 4 + 2] = w_tf; bpp[cw * 4 + 3] = w_tl; } cw = wn; w_ql = i; w_tl = j; }
                            w_qf = i; w_tf = j;
                        }
                        --i; --j;
                    } else { state = src; emit = 0; }
                } else if (state == 1) { if (opp && lane == 0) opp[oc++] = 3; if (!((v >> 2) & 1)) state = 0; --j; }
                else { if (opp && lane == 0) opp[oc++] = 2; if (!((v >> 3) & 1)) state = 0; --i; }
                if (emit) { win = (win << 1) | (uint64_t)bit; nm += bit; ++cols; if (cols >= K) region += ((int)__popcll(win & kmask) >= mid); }
            }
            if (lane == 0 && bpp && cw >= 0 && cw < J.bp_windows) { bpp[cw * 4 + 0] = w_qf; bpp[cw * 4 + 1] = w_ql; bpp[cw * 4 + 2] = w_tf; bpp[cw * 4 + 3] = w_tl; }
            if (opp && lane == 0) { for (int x = i; x >= 0; --x) opp[oc++] = 2; for (int x = j; x >= 0; --x) opp[oc++] = 3; }
            {   // leading end gaps
                const int z = (i + 1) + (j + 1);
                const int zl = z < K ? z : K;
                for (int x = 0; x < zl; ++x) { win <<= 1; ++cols; if (cols >= K) region += ((int)__popcll(win & kmask) >= mid); }
                if (z > zl) { region += (0 >= mid) ? (z - zl) : 0; cols += z - zl; }
            }
            if (cols < K) region = (nm >= mid) ? 1 : 0;      // a single, shorter window (cluster.py:148-154)
            if (lane == 0) {
                if (J.score) J.score[p] = best;
                if (J.ncols) J.ncols[p] = cols;
                if (J.nmatch) J.nmatch[p] = nm;
                if (J.region) J.region[p] = region;
                if (J.span) { J.span[p * 4 + 0] = q_beg; J.span[p * 4 + 1] = q_end; J.span[p * 4 + 2] = t_beg; J.span[p * 4 + 3] = t_end; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int RPL>
static int32_t launch_rpl(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, bool long_class = false)
{
    // long_class: the pairs above the int16 lengths of a partitioned batch - own scratch (the class launches of the same call are still running on theirs), own queue counter
    DevBuf<uint64_t>& TB = long_class ? ctx->tb_long : ctx->tb; DevBuf<int32_t>& BND = long_class ? ctx->bnd_long : ctx->bnd;
    const uint32_t ctr_slot = long_class ? 6u : 0u;
    const uint64_t strip = 64ull * RPL;
    const uint64_t nstrips = (max_qlen + strip - 1) / strip;
    const uint64_t words = (nstrips ? nstrips : 1) * ((uint64_t)max_tlen + 63) * 64;
    uint64_t want = job.npairs;
    const uint64_t by_mem = ctx->scratch_budget / (words * 8 + 1);
    if (want > by_mem) want = by_mem;
    if (want < 1) want = 1;
    const uint32_t seq_lds = ((max_tlen > max_qlen ? max_tlen : max_qlen) + 15u) & ~15u;
    const uint32_t lds_per_wave = 2 * seq_lds + 4096;
    int wpb = 4;
    while (wpb > 1 && (uint64_t)wpb * lds_per_wave > 40 * 1024) wpb >>= 1;
    if ((size_t)wpb * lds_per_wave > 160 * 1024) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "aligner needs %zu bytes of LDS per wave", (size_t)wpb * lds_per_wave);
    if ((size_t)wpb * lds_per_wave > 64 * 1024) HIPCHK(ctx, hipFuncSetAttribute((const void*)k_sg_align<RPL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)wpb * lds_per_wave)));   // reads above 30 k bases: both sequences of a pair in LDS are more than the default limit
    int occ = 0;
    HIPCHK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sg_align<RPL>, 64 * wpb, (size_t)wpb * lds_per_wave));
    if (occ < 1) occ = 1;
    const uint64_t resident = (uint64_t)occ * ctx->n_cu * wpb;          // waves that fit on the chip at once
    if (want > resident) want = resident;
    const uint64_t blocks = (want + wpb - 1) / wpb;
    const uint64_t nwaves = blocks * wpb;
    if (ctx->aln_ctr.n < 16) HIPCHK(ctx, ctx->aln_ctr.alloc(16));
    HIPCHK(ctx, hipMemsetAsync(ctx->aln_ctr.p + ctr_slot, 0, sizeof(uint32_t), ctx->stream));
    const uint32_t bnd_stride = (max_tlen + 15u) & ~15u;
    if (TB.n < nwaves * words) HIPCHK(ctx, TB.alloc(nwaves * words));
    if (BND.n < nwaves * 2ull * bnd_stride) HIPCHK(ctx, BND.alloc(nwaves * 2ull * bnd_stride));
    { ProfScope ps_(ctx, "k_sg_align"); hipLaunchKernelGGL((k_sg_align<RPL>), dim3((unsigned)blocks), dim3(64 * wpb), (size_t)wpb * lds_per_wave, ctx->stream,
                       job, TB.p, words, BND.p, bnd_stride, lds_per_wave, ctx->aln_ctr.p + ctr_slot); }
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}

// measurement (profiling on): DP cells of the call = sum of n x m over its pairs, one atomic per workgroup
__global__ __launch_bounds__(256) void k_align_cells(AlignJob J, unsigned long long* __restrict__ stat)
{
    const uint64_t np = J.npairs_dev ? (uint64_t)*J.npairs_dev : J.npairs;
    unsigned long long c = 0;
    for (uint64_t x = (uint64_t)blockIdx.x * 256 + threadIdx.x; x < np; x += (uint64_t)gridDim.x * 256) {
        const uint64_t p = J.pair_list ? J.pair_list[x] : x; const uint32_t qi = J.qidx[p], ti = J.tidx[p];
        c += (unsigned long long)(J.qoff[qi + 1] - J.qoff[qi]) * (unsigned long long)(J.toff[ti + 1] - J.toff[ti]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(stat, part[0] + part[1] + part[2] + part[3]);
}

int32_t ngsid_launch_align(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, int max_open, uint32_t min_qlen)
{
    if (job.npairs == 0) return NGSID_OK;
    if (ctx->prof && ctx->stat.p && !job.bp) { hipLaunchKernelGGL(k_align_cells, dim3((unsigned)std::min<uint64_t>(1024, (job.npairs + 255) / 256)), dim3(256), 0, ctx->stream, job, ctx->stat.p + 1); HIPCHK(ctx, hipGetLastError()); }
    if (job.npairs > 0xf0000000ull) NGSID_FAIL(ctx, NGSID_ERR_ARG, "more than 2^32 pairs in one aligner call");
    if (max_tlen > NGSID_MAX_READ_LEN || max_qlen > NGSID_MAX_READ_LEN) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "sequence longer than %d in aligner", NGSID_MAX_READ_LEN);
    if (job.k > 64) NGSID_FAIL(ctx, NGSID_ERR_ARG, "window k > 64 unsupported");
    if (!job.ops && !ngsid_opt(ctx, "align32", 0)) {
        if (ngsid_align16_applicable(job, max_qlen, max_tlen, max_open)) return ngsid_launch_align16(ctx, job, max_qlen, max_tlen, min_qlen);   // packed int16 path (bit-identical)
        // Round 5 (reads up to 65 535 bases): a large batch with a few long sequences keeps its short pairs in the int16 instances - the pairs with a query or a
        // target above NGSID_ALIGN16_MAXLEN become a class of their own that runs here, in int32, after the others (without this one 40 kb read among the
        // representatives would send a million 750-base pairs through the int32 kernel at one wave per CU)
        const uint32_t q16 = std::min<uint32_t>(max_qlen, NGSID_ALIGN16_MAXLEN), t16 = std::min<uint32_t>(max_tlen, NGSID_ALIGN16_MAXLEN);
        if (job.npairs >= 4096 && q16 > 256 && !job.pair_list && !ngsid_opt(ctx, "align_noclass", 0) && ngsid_align16_applicable(job, q16, t16, max_open)) {      // (= the conditions of the class path of ngsid_launch_align16)
            int32_t rc = ngsid_launch_align16(ctx, job, q16, t16, min_qlen, NGSID_ALIGN16_MAXLEN); if (rc) return rc;
            AlignJob jl = job; jl.pair_list = ctx->aln_cls.p + (size_t)NGSID_ALIGN_LONG_CLASS * job.npairs; jl.npairs_dev = ctx->aln_ctr.p + 8 + NGSID_ALIGN_LONG_CLASS;
            return launch_rpl<16>(ctx, jl, max_qlen, max_tlen, true);
        }
    }
    if (max_qlen <= 256) return launch_rpl<4>(ctx, job, max_qlen, max_tlen);
    if (max_qlen <= 512) return launch_rpl<8>(ctx, job, max_qlen, max_tlen);
    if (max_qlen <= 768) return launch_rpl<12>(ctx, job, max_qlen, max_tlen);
    if (max_qlen <= 832) return launch_rpl<13>(ctx, job, max_qlen, max_tlen);     // tight fit: fewer idle lanes and rows per step
    if (max_qlen <= 896) return launch_rpl<14>(ctx, job, max_qlen, max_tlen);
    return launch_rpl<16>(ctx, job, max_qlen, max_tlen);
}
