// k_minimizers.hip - (a1-a3) homopolymer compression, minimizer extraction, HPC / raw error rates.
//
// Replaces cluster.py:265 (HPC), :16-39 (get_kmer_minimizers), :279-291 (HPC quality + error rate) and the
// raw-quality mean of :185-188.  One 256-thread workgroup per read; the HPC string and the k-mer codes of
// the read are staged in LDS (9 bytes per base), window minima are taken from LDS, and the (code,pos)
// pairs are written compacted to HBM at the read's own base offset (a read of n bases has at most n-k+1
// minimizers), so no allocation pass or atomics are needed and the layout is deterministic.
// HBM traffic per read: 2L in (bases + qualities, coalesced), 12 B per minimizer out.
#include "ngsid_internal.h"
#include "../../include/ngsid_tables.h"

__constant__ double c_phred_p[128];
static bool g_tables_loaded[16] = {false};

__device__ __forceinline__ int enc3(uint8_t c) {
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 3; case 'N': return 4; case 'T': return 5; default: return -1; }
}

// exclusive prefix sum of one int per thread over a 256-thread block; returns the exclusive value, *total = block sum
__device__ __forceinline__ int block_excl_scan256(int v, int* lds_w /*4 ints*/, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == 63) lds_w[wv] = x;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base += lds_w[i];
    *total = lds_w[0] + lds_w[1] + lds_w[2] + lds_w[3];
    __syncthreads();
    return base + x - v;
}

extern "C" __global__ __launch_bounds__(256)
void k_hpc_minimizers(const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, const uint64_t* __restrict__ off, uint64_t nreads,
                      int k, int w, uint64_t* __restrict__ out_codes, uint32_t* __restrict__ out_pos, uint32_t* __restrict__ out_cnt,
                      uint32_t* __restrict__ out_hlen, double* __restrict__ out_herr, double* __restrict__ out_rawerr, int* __restrict__ flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint64_t r = blockIdx.x;
    if (r >= nreads) return;
    const uint64_t base = off[r];
    const int n = (int)(off[r + 1] - base);
    const int tid = threadIdx.x;
    // LDS carve: [0,1024) hist_hpc(128 int) + hist_raw(128 int); [1024,1024+64) scan scratch + misc; then codes (8B aligned), then hs
    int* hist_h = (int*)smem; int* hist_r = hist_h + 128;
    int* scr = hist_r + 128;                 // 16 ints
    int* am = scr + 16;                      // 257 ints (window argmin exchange), padded to 272
    double* term_h = (double*)(smem + 1024 + 64 + 272 * 4); double* term_r = term_h + 128;      // per quality character: count x error probability
    uint64_t* codes = (uint64_t*)(smem + 1024 + 64 + 272 * 4 + 2048);
    const int W = w - k + 1;
    const int ncodes_cap = (n > W ? n : W) + 1;
    uint8_t* hs = (uint8_t*)(codes + ncodes_cap);
    const uint8_t* s = seq + base; const uint8_t* q = qual ? qual + base : nullptr;

    if (tid < 128) { hist_h[tid] = 0; hist_r[tid] = 0; }
    __syncthreads();

    // ---- phase 1: run heads, HPC string, best quality per run, histograms
    const int CH = (n + 255) / 256;
    const int i0 = tid * CH, i1 = min(n, i0 + CH);
    int heads = 0;
    for (int i = i0; i < i1; ++i) heads += (i == 0 || s[i] != s[i - 1]);
    int total_heads;
    int hidx = block_excl_scan256(heads, scr, &total_heads);
    for (int i = i0; i < i1; ++i) {
        if (q) atomicAdd(&hist_r[q[i] & 127], 1);
        if (i == 0 || s[i] != s[i - 1]) {
            const uint8_t c = s[i];
            hs[hidx] = (uint8_t)enc3(c);          // 3-bit letter code, 0xff = outside ACGTN (reported in phase 2)
            if (q) {
                uint8_t best = q[i];
                for (int j = i + 1; j < n && s[j] == c; ++j) if (c_phred_p[q[j] & 127] < c_phred_p[best & 127]) best = q[j];
                atomicAdd(&hist_h[best & 127], 1);
            }
            ++hidx;
        }
    }
    __syncthreads();
    const int hl = total_heads;
    // error rates = sum over the quality characters in ascending order of count x p (the order fixes the rounding: ngsid_oracle.c does the same);
    // the products are formed by 128 threads, the two ordered sums run on two different waves
    if (tid < 128) { term_h[tid] = (double)hist_h[tid] * c_phred_p[tid]; term_r[tid] = (double)hist_r[tid] * c_phred_p[tid]; }
    __syncthreads();
    const double qnan = __longlong_as_double(0x7ff8000000000000ULL);
    if (tid == 0) {
        out_hlen[r] = (uint32_t)hl;
        double sh = 0.0;
        if (q) { for (int c = 0; c < 128; ++c) sh = sh + term_h[c]; }
        out_herr[r] = (q && hl > 0) ? sh / (double)hl : qnan;
    } else if (tid == 64) {
        double sr = 0.0;
        if (q) { for (int c = 0; c < 128; ++c) sr = sr + term_r[c]; }
        out_rawerr[r] = (q && n > 0) ? sr / (double)n : qnan;
    }
    if (hl < k) { if (tid == 0) out_cnt[r] = 0; return; }

    // ---- phase 2: k-mer codes (3 bits/base, left aligned, zero padded past the end)
    const int nk = hl - k + 1;
    const int nc = nk > W ? nk : W;
    int bad = 0;
    for (int i = tid; i < nc; i += 256) {
        uint64_t c = 0;
        for (int t = 0; t < k; ++t) {
            int e = 0;
            if (i + t < hl) { e = hs[i + t]; if (e == 0xff) { bad = 1; e = 0; } }
            c = (c << 3) | (uint64_t)e;
        }
        codes[i] = c;
    }
    if (bad) atomicExch(flag, 1 + (int)(r & 0x3fffffff));
    __syncthreads();

    // ---- phase 3: leftmost window minima, emit on position change (ordered compaction, 256 windows per round)
    const int nwin = nk >= W ? nk - W + 1 : 1;
    int emitted = 0;
    if (tid == 0) am[0] = -1;
    for (int s0 = 0; s0 < nwin; s0 += 256) {
        const int sidx = s0 + tid;
        int best = -1;
        if (sidx < nwin) {
            best = sidx; uint64_t bc = codes[sidx];
            for (int j = sidx + 1; j < sidx + W; ++j) { const uint64_t c = codes[j]; if (c < bc) { bc = c; best = j; } }
        }
        __syncthreads();                 // am[0] from the previous round is in place
        am[tid + 1] = best;
        __syncthreads();
        const int prev = am[tid];
        const int f = (sidx < nwin) && (best != prev);
        int tot;
        const int ex = block_excl_scan256(f, scr, &tot);
        if (f) { out_codes[base + emitted + ex] = codes[best]; out_pos[base + emitted + ex] = (uint32_t)best; }
        emitted += tot;
        const int last = am[256];
        __syncthreads();
        if (tid == 0) am[0] = last;
    }
    if (tid == 0) out_cnt[r] = (uint32_t)emitted;
}

int32_t ngsid_launch_minimizers(ngsid_ctx* ctx, const DevReads& R, int k, int w,
                                uint64_t* d_codes, uint32_t* d_pos, uint32_t* d_cnt, uint32_t* d_hlen, double* d_herr, double* d_rawerr, int* d_flag)
{
    if (k < 1 || k > NGSID_MAX_K || w < k || w > 255) NGSID_FAIL(ctx, NGSID_ERR_ARG, "k must be in [1,%d] and k <= w <= 255 (k=%d w=%d)", NGSID_MAX_K, k, w);
    if (R.maxlen > NGSID_MAX_READ_LEN) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "read of %u bases exceeds NGSID_MAX_READ_LEN=%d", R.maxlen, NGSID_MAX_READ_LEN);
    if (!g_tables_loaded[ctx->device & 15]) {
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_phred_p), NGSID_PHRED_P, sizeof(double) * 128));
        g_tables_loaded[ctx->device & 15] = true;
    }
    if (R.n == 0) return NGSID_OK;
    const int W = w - k + 1;
    const size_t ncap = (size_t)((int)R.maxlen > W ? (int)R.maxlen : W) + 1;
    size_t lds = 1024 + 64 + 272 * 4 + 2048 + ncap * 8 + (size_t)R.maxlen + 16;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 64 * 1024) HIPCHK(ctx, hipFuncSetAttribute((const void*)k_hpc_minimizers, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    { ProfScope ps_(ctx, "k_hpc_minimizers"); hipLaunchKernelGGL(k_hpc_minimizers, dim3((unsigned)R.n), dim3(256), lds, ctx->stream,
                       R.seq, R.qual, R.off, R.n, k, w, d_codes, d_pos, d_cnt, d_hlen, d_herr, d_rawerr, d_flag); }
    HIPCHK(ctx, hipGetLastError());
    return NGSID_OK;
}
