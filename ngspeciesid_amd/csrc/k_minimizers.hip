// k_minimizers.hip - (a1-a3) homopolymer compression, minimizer extraction, HPC / raw error rates.
//
// Replaces cluster.py:265 (HPC), :16-39 (get_kmer_minimizers), :279-291 (HPC quality + error rate) and the raw-quality mean of :185-188.
// ONE WAVE PER READ (round 2; round 1 used a 256-thread workgroup with __syncthreads phases, 9.9 ms per million reads):
//   1. bases + qualities are staged in LDS with coalesced dword loads (2 bytes per base - the only HBM reads of the kernel);
//   2. run heads by ballot, HPC letters by prefix popcount, best quality of a run by its head lane (runs are short), quality histograms by
//      LDS atomics; the two error-rate sums run on two lanes over the quality characters that occur, in ascending character code, as ordered
//      FP64 sums (same rounding as the oracle: skipped terms are exact zeros);
//   3. k-mer codes (3 bits per letter, left aligned, zero padded past the end) from aligned dword reads of the HPC string;
//   4. window minima through a sparse table of argmin positions (log2(w-k+1) doubling passes instead of a scan of w-k+1 codes per window),
//      emission on position change by ballot-ordered compaction, (code, pos) pairs written at the read's own base offset in HBM
//      (a read of n bases has at most n-k+1 minimizers: no allocation pass, no atomics, deterministic layout).
// k <= 21 fits one 64-bit code.  22 <= k <= 42 (KW = 2) keeps (hi, lo) pairs - lo = the last 21 letters - and a rename pass afterwards
// replaces them by their dense rank over the whole launch (order preserving, equal k-mers equal code), so everything downstream still
// sees 64-bit order-preserving codes.  HBM traffic per read: 2L in, 12 B per minimizer out.
#include "ngsid_internal.h"
#include "../../include/ngsid_tables.h"
#include <hipcub/hipcub.hpp>

__constant__ double c_phred_p[128];
static bool g_tables_loaded[16] = {false};

#define MZ_LDSP __attribute__((address_space(3)))

__device__ __forceinline__ int enc3(uint8_t c) {
    switch (c) { case 'A': return 1; case 'C': return 2; case 'G': return 3; case 'N': return 4; case 'T': return 5; default: return -1; }
}
// four letters (one per byte, values 0..5) -> 12 bits, first letter in the most significant position
__device__ __forceinline__ unsigned pack4(unsigned w) { return ((w & 7u) << 9) | (((w >> 8) & 7u) << 6) | (((w >> 16) & 7u) << 3) | ((w >> 24) & 7u); }

struct Code2 { uint64_t hi, lo; };
template <int KW> struct CodeT;
template <> struct CodeT<1> { uint64_t lo; __device__ __forceinline__ bool less(const CodeT<1>& o) const { return lo < o.lo; } };
template <> struct CodeT<2> { uint64_t hi, lo; __device__ __forceinline__ bool less(const CodeT<2>& o) const { return hi < o.hi || (hi == o.hi && lo < o.lo); } };

// LDS of one wave (bytes): hs [NC + 64] | am [NC u16] | then EITHER sraw [ML4] | qraw [ML4] | hist_h, hist_r [2 x 128 int] (phases 1-2) OR lo [NC u64] | hi [NC u64, KW = 2] (phases 3-4)
// LEAN (reads whose full layout would not fit the 160 KB of a CU: above ~14 800 bases at k <= 21, ~8 600 at k > 21): no code arrays - every
// k-mer code is rebuilt from the HPC letters where it is compared or emitted (k / 4 dword reads each), 5 bytes per base instead of 11 / 19.
// REG (round 3; windows of up to 16 k-mers, k <= 21 - the ONT default k13 / w20 has 8): no code array and no argmin array either.  The window minima are formed in REGISTERS:
// 64 consecutive k-mer codes, one per lane, rebuilt from the HPC letters; log2(window) doubling steps with lane shifts (the leftmost-minimum sparse table of the stored layout,
// without the table); a chunk yields 64 - (window - 1) windows.  3.5 KB of LDS per 800-base read instead of 8.9 KB: the waves per CU are no longer bound by LDS.
// LONG (round 5: reads above MZ_SHORT_LEN = 16 384 and up to NGSID_MAX_READ_LEN = 65 535 bases): the raw read is NOT staged - phases 1-2 read bases and qualities from HBM -, LDS holds
// the HPC letters and the two histograms only (65 KB at 65 535 bases), and every window scans its k-mers directly (codes rebuilt from the HPC letters).  Such reads are rare in amplicon
// data; they get a launch of their own (the other reads keep their layout), one wave per read.
__host__ __device__ inline size_t mz_lds_per_wave(uint32_t maxlen, int W, int KW, int mode = 0 /* 0 stored, 1 lean, 2 reg, 3 long */)
{
    const size_t ML4 = ((size_t)maxlen + 7) & ~(size_t)3;
    const size_t NC = (((size_t)maxlen > (size_t)W ? (size_t)maxlen : (size_t)W) + 4 + 3) & ~(size_t)3;
    if (mode == 3) return (NC + 64) + 1024;
    const size_t phase12 = 2 * ML4 + 1024, phase34 = mode ? 0 : NC * 8 * (size_t)KW;      // staged read + histograms | k-mer codes: never live together
    return (NC + 64) + (mode == 2 ? 0 : NC * 2) + (phase12 > phase34 ? phase12 : phase34);
}

template <int KW, int MODE>
__global__ __launch_bounds__(256)
void k_hpc_minimizers(const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, const uint64_t* __restrict__ off, uint64_t nreads,
                      int k, int w, uint32_t maxlen, uint32_t lds_per_wave,
                      uint64_t* __restrict__ out_codes, uint64_t* __restrict__ out_hi, uint32_t* __restrict__ out_pos, uint32_t* __restrict__ out_cnt,
                      uint32_t* __restrict__ out_hlen, double* __restrict__ out_herr, double* __restrict__ out_rawerr, int* __restrict__ flag, uint64_t out_base, uint32_t skip_le)
{
    // skip_le: reads of at most that many bases belong to another launch of the same call (0: none); reads above maxlen (the layout bound) likewise
    // out_base: base offset of the first read of this launch - the sparse outputs (out_codes / out_hi / out_pos) start there (round 5: a launch covers one chunk of reads
    // and writes into a scratch of that chunk's size)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool LEAN = MODE == 1, REG = MODE == 2, LONG = MODE == 3;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    if (r >= nreads) return;
    const int W = w - k + 1;
    const size_t ML4 = ((size_t)maxlen + 7) & ~(size_t)3;
    const size_t NC = (((size_t)maxlen > (size_t)W ? (size_t)maxlen : (size_t)W) + 4 + 3) & ~(size_t)3;
    unsigned char* base_l = smem + (size_t)wv * lds_per_wave;
    uint8_t* hs = base_l; uint16_t* am = (uint16_t*)(hs + NC + 64);
    unsigned char* un = (unsigned char*)am + (REG ? 0 : NC * 2);                          // 8-aligned: NC is a multiple of 4, the wave slice of 16 (REG: no argmin array)
    const uint64_t base = off[r];
    const int n = (int)(off[r + 1] - base);
    if ((uint32_t)n > maxlen || (skip_le && (uint32_t)n <= skip_le)) return;
    const uint8_t* s = seq + base; const uint8_t* q = qual ? qual + base : nullptr;
    const uint8_t* sraw = LONG ? s : un; const uint8_t* qraw = LONG ? q : un + ML4;                      // LONG: the read stays in HBM
    int* hist_h = LONG ? (int*)(hs + NC + 64) : (int*)(un + 2 * ML4); int* hist_r = hist_h + 128;
    uint64_t* clo = (uint64_t*)un; uint64_t* chi = clo + NC;
    // wave-private LDS: instructions of one wave execute in order, so a wave barrier + a drained LDS queue orders its own traffic
    auto lsync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };

    // ---- 1. stage bases / qualities (dword loads where the read's start allows it), clear the histograms
    for (int c = lane; c < 256; c += 64) hist_h[c] = 0;
    if constexpr (!LONG) {
        // the LDS copy starts at the same offset modulo 4 as the global address, so the body moves as aligned dwords on both sides
        auto stage = [&](const uint8_t* g, uint8_t* l) {
            const int head = (int)((4 - ((uintptr_t)g & 3)) & 3);          // bytes until the global address is dword aligned
            for (int i = lane; i < head && i < n; i += 64) l[i] = g[i];
            const int nd = n > head ? (n - head) >> 2 : 0;
            for (int d0 = 0; d0 < nd; d0 += 256) {                                    // four dwords per lane in flight: one HBM round trip per KB
                unsigned v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int d = d0 + 64 * u + lane; v4[u] = d < nd ? *(const unsigned*)(g + head + 4 * d) : 0u; }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int d = d0 + 64 * u + lane; if (d < nd) *(unsigned*)(l + head + 4 * d) = v4[u]; }
            }
            for (int i = head + 4 * nd + lane; i < n; i += 64) l[i] = g[i];
        };
        sraw += (uintptr_t)s & 3; stage(s, (uint8_t*)sraw);
        if (q) { qraw += (uintptr_t)q & 3; stage(q, (uint8_t*)qraw); }
    }
    lsync();

    // ---- 2. run heads, HPC letters, best quality per run, histograms
    int hl = 0; int bad = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int i = c0 + lane; const bool in = i < n;
        const uint8_t ch = in ? sraw[i] : 0;
        const bool head = in && (i == 0 || ch != sraw[i - 1]);
        const unsigned long long m = __ballot(head);
        const int idx = hl + __popcll(m & ((1ull << lane) - 1ull));
        if (in && q) atomicAdd(&hist_r[qraw[i] & 127], 1);
        if (head) {
            const int e = enc3(ch); if (e < 0) bad = 1;
            hs[idx] = (uint8_t)(e < 0 ? 0 : e);
            if (q) {
                // the quality with the LOWEST error probability, first on ties (cluster.py:283).  p(q) = min(10^(-(q-33)/10), 0.79433) is constant up to
                // '!' (33) and strictly decreasing above it (checked on the table when it is loaded), so p(a) < p(b) <=> max(a,33) > max(b,33)
                uint8_t best = qraw[i]; int bk = max((int)(best & 127), 33);
                for (int j = i + 1; j < n && sraw[j] == ch; ++j) { const int kq = max((int)(qraw[j] & 127), 33); if (kq > bk) { bk = kq; best = qraw[j]; } }
                atomicAdd(&hist_h[best & 127], 1);
            }
        }
        hl += __popcll(m);
    }
    if (__ballot(bad)) { if (lane == 0) atomicExch(flag, 1 + (int)(r & 0x3fffffff)); }
    for (int i = hl + lane; i < (int)(NC + 64); i += 64) hs[i] = 0;                       // zero padding past the end: the "end" symbol of truncated k-mers, and room for the dword reads
    lsync();
    // error rates: sum over the quality characters in ascending code of count x p, as ORDERED FP64 sums (zero terms are exact and skipped).  The
    // products are formed by all lanes (lane c: characters c and 64 + c, table entries in registers), the two sums run side by side in lane 0
    // (HPC string) and lane 1 (raw read), every term broadcast from its lane by v_readlane
    {
        const double qnan = __longlong_as_double(0x7ff8000000000000ULL);
        double acc = 0.0;
        if (q) for (int half = 0; half < 2; ++half) {
            const int c = half * 64 + lane;
            const int nh = hist_h[c], nr = hist_r[c]; const double pc = c_phred_p[c];
            const double th = (double)nh * pc, tr = (double)nr * pc;
            unsigned long long occ = __ballot(nh != 0 || nr != 0);
            const unsigned thl = (unsigned)__double_as_longlong(th), thh = (unsigned)(__double_as_longlong(th) >> 32), trl = (unsigned)__double_as_longlong(tr), trh = (unsigned)(__double_as_longlong(tr) >> 32);
            while (occ) {
                const int b = __builtin_ctzll(occ); occ &= occ - 1;
                const unsigned a0 = __builtin_amdgcn_readlane(thl, b), a1 = __builtin_amdgcn_readlane(thh, b), b0 = __builtin_amdgcn_readlane(trl, b), b1 = __builtin_amdgcn_readlane(trh, b);
                const double t = __longlong_as_double((long long)(((unsigned long long)(lane == 0 ? a1 : b1) << 32) | (lane == 0 ? a0 : b0)));
                acc = acc + t;
            }
        }
        if (lane == 0) { out_hlen[r] = (uint32_t)hl; out_herr[r] = (q && hl > 0) ? acc / (double)hl : qnan; }
        else if (lane == 1) out_rawerr[r] = (q && n > 0) ? acc / (double)n : qnan;
    }
    if (hl < k) { if (lane == 0) out_cnt[r] = 0; return; }

    // ---- 3. k-mer codes: letters i .. i+k-1 of the HPC string, 3 bits each, first letter most significant, zeros past the end
    const int nk = hl - k + 1;
    const int nc = nk > W ? nk : W;
    const int nfull = k >> 2, rem = k & 3;
    auto kmer_code = [&](int i, uint64_t& lo, uint64_t& hi) {
        const unsigned* hw = (const unsigned*)(hs + (i & ~3)); const int sh = i & 3;
        lo = 0; hi = 0; unsigned prev = hw[0];
        for (int t = 0; t <= nfull; ++t) {
            const unsigned next = hw[t + 1];
            unsigned wq = __builtin_amdgcn_alignbyte(next, prev, sh);           // bytes i+4t .. i+4t+3
            prev = next;
            unsigned bits = 12; unsigned pk = pack4(wq);
            if (t == nfull) { if (!rem) break; bits = 3u * (unsigned)rem; pk >>= (12u - bits); }
            if (KW == 2) hi = (hi << bits) | (lo >> (63 - bits));
            lo = ((lo << bits) | pk) & 0x7fffffffffffffffULL;
        }
    };
    const int nwin = nk >= W ? nk - W + 1 : 1;
    if constexpr (LONG) {
        // ---- 3 + 4 directly: lane l of a round owns window s0 + l and scans its k-mers left to right (leftmost minimum: a later k-mer wins only if strictly smaller)
        int emitted = 0, carry = -1;
        for (int s0 = 0; s0 < nwin; s0 += 64) {
            const int sidx = s0 + lane; int best = -1; CodeT<KW> cb; cb.lo = 0; if constexpr (KW == 2) cb.hi = 0;
            if (sidx < nwin) {
                const int e = min(sidx + W, nc);
                uint64_t l, h; kmer_code(sidx, l, h); cb.lo = l; if constexpr (KW == 2) cb.hi = h; best = sidx;
                for (int j = sidx + 1; j < e; ++j) {
                    CodeT<KW> cj; kmer_code(j, l, h); cj.lo = l; if constexpr (KW == 2) cj.hi = h;
                    if (cj.less(cb)) { cb = cj; best = j; }
                }
            }
            int prev = __shfl_up(best, 1); if (lane == 0) prev = carry;
            const bool f = sidx < nwin && best != prev;
            const unsigned long long m = __ballot(f);
            if (f) {
                const uint64_t o = base - out_base + (uint64_t)(emitted + __popcll(m & ((1ull << lane) - 1ull)));
                out_codes[o] = cb.lo; if constexpr (KW == 2) out_hi[o] = cb.hi;
                out_pos[o] = (uint32_t)best;
            }
            emitted += __popcll(m);
            const int nv = min(64, nwin - s0); carry = __shfl(best, nv - 1);
        }
        if (lane == 0) out_cnt[r] = (uint32_t)emitted;
        return;
    }
    if constexpr (REG) {
        // ---- 3 + 4 in registers (KW == 1, W <= 16): lane l of a chunk holds the code of k-mer b + l (positions past nc: +infinity); after the doubling steps it holds the
        //      leftmost minimum of [b + l, b + l + W) - valid for the first 64 - (W - 1) lanes, which are the windows of this chunk.  Ties keep the left candidate.
        const int S = 64 - (W - 1);
        int emitted = 0, carry = -1;
        for (int b = 0; b < nwin; b += S) {
            const int p = b + lane;
            uint64_t code = ~0ull; int pos = p;
            if (p < nc) { uint64_t lo, hi; kmer_code(p, lo, hi); code = lo; }
            int span = 1;
            while (span * 2 <= W) {
                const uint64_t oc = __shfl_down((unsigned long long)code, span); const int op = __shfl_down(pos, span);
                if (lane + span < 64 && oc < code) { code = oc; pos = op; }
                span *= 2;
            }
            if (W > span) {
                const int d = W - span;
                const uint64_t oc = __shfl_down((unsigned long long)code, d); const int op = __shfl_down(pos, d);
                if (lane + d < 64 && oc < code) { code = oc; pos = op; }
            }
            const int sidx = b + lane; const bool valid = lane < S && sidx < nwin;
            const int best = valid ? pos : -1;
            int prev = __shfl_up(best, 1); if (lane == 0) prev = carry;
            const bool f = valid && best != prev;
            const unsigned long long m = __ballot(f);
            if (f) {
                const uint64_t o = base - out_base + (uint64_t)(emitted + __popcll(m & ((1ull << lane) - 1ull)));
                out_codes[o] = code; out_pos[o] = (uint32_t)best;
            }
            emitted += __popcll(m);
            const int nv = min(S, nwin - b); carry = __shfl(best, nv - 1);
        }
        if (lane == 0) out_cnt[r] = (uint32_t)emitted;
        return;
    }
    for (int i = lane; i < nc; i += 64) {
        // positions past the end of the string hold zeros already (hs padding); letters beyond hl inside a k-mer are zeros = "end" symbol
        if (!LEAN) { uint64_t lo, hi; kmer_code(i, lo, hi); clo[i] = lo; if (KW == 2) chi[i] = hi; }
        am[i] = (uint16_t)i;
    }
    lsync();

    // ---- 4. sparse table of argmin positions: after pass p, am[i] = leftmost minimum of codes[i .. i+2^p) (clipped at nc)
    auto better = [&](int a, int b) -> int {           // leftmost minimum of two candidates, a's range starts left of b's
        CodeT<KW> ca, cb;
        if (LEAN) { uint64_t l, h; kmer_code(a, l, h); ca.lo = l; if constexpr (KW == 2) ca.hi = h; kmer_code(b, l, h); cb.lo = l; if constexpr (KW == 2) cb.hi = h; }
        else { ca.lo = clo[a]; cb.lo = clo[b]; if constexpr (KW == 2) { ca.hi = chi[a]; cb.hi = chi[b]; } }
        return cb.less(ca) ? b : a;
    };
    int span = 1;
    while (span * 2 <= W) {
        for (int i0 = 0; i0 < nc; i0 += 64) {
            const int i = i0 + lane; int v = 0;
            if (i < nc) { v = am[i]; if (i + span < nc) v = better(v, am[i + span]); }
            lsync();                                    // every lane of the chunk has read before any writes (in place, ascending chunks)
            if (i < nc) am[i] = (uint16_t)v;
        }
        lsync();
        span *= 2;
    }
    // leftmost window minima, emit on position change (ordered compaction, 64 windows per round)
    int emitted = 0, carry = -1;
    for (int s0 = 0; s0 < nwin; s0 += 64) {
        const int sidx = s0 + lane; int best = -1;
        if (sidx < nwin) { best = am[sidx]; if (W > span) best = better(best, am[sidx + W - span]); }
        int prev = __shfl_up(best, 1); if (lane == 0) prev = carry;
        const bool f = sidx < nwin && best != prev;
        const unsigned long long m = __ballot(f);
        if (f) {
            const uint64_t o = base - out_base + (uint64_t)(emitted + __popcll(m & ((1ull << lane) - 1ull)));
            if (LEAN) { uint64_t l, h; kmer_code(best, l, h); out_codes[o] = l; if (KW == 2) out_hi[o] = h; }
            else { out_codes[o] = clo[best]; if (KW == 2) out_hi[o] = chi[best]; }
            out_pos[o] = (uint32_t)best;
        }
        emitted += __popcll(m);
        const int nv = min(64, nwin - s0); carry = __shfl(best, nv - 1);
    }
    if (lane == 0) out_cnt[r] = (uint32_t)emitted;
}

// ---------------------------------------------------------------------------------------------- rename pass for k > 21
namespace {
__global__ void k_iota32(uint32_t* __restrict__ idx, uint64_t n)
{ const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) idx[i] = (uint32_t)i; }
__global__ void k_gather64(const uint64_t* __restrict__ src, const uint32_t* __restrict__ idx, uint64_t n, uint64_t* __restrict__ dst)
{ const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) dst[i] = src[idx[i]]; }
__global__ void k_mz_flags(const uint64_t* __restrict__ shi, const uint64_t* __restrict__ lo_c, const uint32_t* __restrict__ perm, uint64_t n, uint32_t* __restrict__ fl)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    fl[i] = (i == 0) ? 0u : ((shi[i] != shi[i - 1] || lo_c[perm[i]] != lo_c[perm[i - 1]]) ? 1u : 0u);
}
__global__ void k_mz_scatter(const uint32_t* __restrict__ rank, const uint32_t* __restrict__ perm, uint64_t n, uint64_t* __restrict__ out_c)
{ const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out_c[perm[i]] = (uint64_t)rank[i]; }
// sparse (at the reads' base offsets, relative to `sbase`) -> compact CSR; one 64-lane workgroup per read
__global__ void k_mz_gather(const uint64_t* __restrict__ roff, const uint32_t* __restrict__ cnt, const uint64_t* __restrict__ moff, uint64_t n, uint64_t sbase,
                            const uint64_t* __restrict__ scodes, const uint64_t* __restrict__ shi, const uint32_t* __restrict__ spos,
                            uint64_t* __restrict__ codes, uint64_t* __restrict__ hi, uint32_t* __restrict__ pos)
{
    const uint64_t r = blockIdx.x; if (r >= n) return;
    const uint64_t src = roff[r] - sbase, dst = moff[r]; const uint32_t c = cnt[r];
    for (uint32_t i = threadIdx.x; i < c; i += blockDim.x) { codes[dst + i] = scodes[src + i]; pos[dst + i] = spos[src + i]; if (hi) hi[dst + i] = shi[src + i]; }
}
}  // namespace

// compact (lo, hi) code pairs -> dense, order-preserving ranks written over d_lo (equal k-mers equal code)
static int32_t mz_rename_wide(ngsid_ctx* ctx, uint64_t* d_lo, const uint64_t* d_hi, uint64_t M)
{
    if (M == 0) return NGSID_OK;
    if (M > 0xfffffff0ull) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "more than 2^32 minimizers in one call with k > 21");
    DevBuf<unsigned char> tmp; size_t tb = 0;
    DevBuf<uint64_t> k2, shi; DevBuf<uint32_t> idx, p1, p2, fl, rk;
    HIPCHK(ctx, k2.alloc(M)); HIPCHK(ctx, shi.alloc(M)); HIPCHK(ctx, idx.alloc(M)); HIPCHK(ctx, p1.alloc(M)); HIPCHK(ctx, p2.alloc(M)); HIPCHK(ctx, fl.alloc(M)); HIPCHK(ctx, rk.alloc(M));
    const unsigned gb = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(k_iota32, dim3(gb), dim3(256), 0, ctx->stream, idx.p, M);
    // LSD: stable sort by lo, then stable sort by hi
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb, d_lo, k2.p, idx.p, p1.p, (int)M, 0, 63, ctx->stream));
    HIPCHK(ctx, tmp.alloc(tb));
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, d_lo, k2.p, idx.p, p1.p, (int)M, 0, 63, ctx->stream));
    hipLaunchKernelGGL(k_gather64, dim3(gb), dim3(256), 0, ctx->stream, d_hi, p1.p, M, k2.p);                        // hi in lo-sorted order
    size_t tb2 = 0; HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, k2.p, shi.p, p1.p, p2.p, (int)M, 0, 64, ctx->stream));
    if (tb2 > tb) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, tmp.alloc(tb2)); tb = tb2; }
    HIPCHK(ctx, hipcub::DeviceRadixSort::SortPairs(tmp.p, tb2, k2.p, shi.p, p1.p, p2.p, (int)M, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_mz_flags, dim3(gb), dim3(256), 0, ctx->stream, shi.p, d_lo, p2.p, M, fl.p);
    size_t tb3 = 0; HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, tb3, fl.p, rk.p, (int)M, ctx->stream));
    if (tb3 > tb) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); HIPCHK(ctx, tmp.alloc(tb3)); tb = tb3; }
    HIPCHK(ctx, hipcub::DeviceScan::InclusiveSum(tmp.p, tb3, fl.p, rk.p, (int)M, ctx->stream));
    hipLaunchKernelGGL(k_mz_scatter, dim3(gb), dim3(256), 0, ctx->stream, rk.p, p2.p, M, k2.p);                      // k2[compact index] = rank
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(d_lo, k2.p, 8 * M, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NGSID_OK;
}

// one launch over the reads of R (a view: R.off may point into a longer offset array); every output pointer is indexed by the view's read number, the
// sparse arrays d_codes / d_hi / d_pos by the reads' base offsets minus out_base (= the base offset of the view's first read: the scratch holds one chunk)
#define MZ_SHORT_LEN 16384u             // reads up to this length run the LDS-staged layouts; longer ones (up to NGSID_MAX_READ_LEN) the LONG layout in a launch of their own
static int32_t mz_launch(ngsid_ctx* ctx, const DevReads& R, int k, int w, uint64_t out_base,
                         uint64_t* d_codes, uint64_t* d_hi, uint32_t* d_pos, uint32_t* d_cnt, uint32_t* d_hlen, double* d_herr, double* d_rawerr, int* d_flag)
{
    if (R.n == 0) return NGSID_OK;
    const int W = w - k + 1;
    const int KW = k <= 21 ? 1 : 2;
    const int wpb = 1;                   // one wave per workgroup: the LDS slice of a read (8.3 KB at 750 bases) is what bounds the waves per CU (measured: 9.5 ms per 10^6 reads, 10.2 ms with four waves per workgroup)
    const unsigned blocks = (unsigned)((R.n + wpb - 1) / wpb);
    auto run = [&](int mode, uint32_t maxlen, uint32_t skip_le) -> int32_t {
        const size_t lpw = (mz_lds_per_wave(maxlen, W, KW, mode) + 15) & ~(size_t)15;
        const size_t lds = lpw * wpb;
        if (lds > 160 * 1024) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "minimizer kernel needs %zu bytes of LDS", lds);
        ProfScope ps_(ctx, "k_hpc_minimizers");
        auto go = [&](auto kern, uint64_t* hi_p) -> hipError_t {
            if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * wpb), lds, ctx->stream, R.seq, R.qual, R.off, R.n, k, w, maxlen, (uint32_t)lpw, d_codes, hi_p, d_pos, d_cnt, d_hlen, d_herr, d_rawerr, d_flag, out_base, skip_le);
            return hipSuccess;
        };
        if (KW == 1) HIPCHK(ctx, mode == 3 ? go(k_hpc_minimizers<1, 3>, nullptr) : mode == 2 ? go(k_hpc_minimizers<1, 2>, nullptr) : mode == 1 ? go(k_hpc_minimizers<1, 1>, nullptr) : go(k_hpc_minimizers<1, 0>, nullptr));
        else HIPCHK(ctx, mode == 3 ? go(k_hpc_minimizers<2, 3>, d_hi) : mode == 1 ? go(k_hpc_minimizers<2, 1>, d_hi) : go(k_hpc_minimizers<2, 0>, d_hi));
        HIPCHK(ctx, hipGetLastError());
        return NGSID_OK;
    };
    // layout: 2 = registers (windows of up to 16 k-mers, one-word codes), 1 = lean (long reads: the stored layout needs 180 KB at 16 384 bases), 0 = stored;
    // ngsid_ctx_option "minimizers_mode" (1 stored, 2 lean, 3 registers, 4 long) / "minimizers_lean" pick one for tests - same output from all
    const long long want = ngsid_opt(ctx, "minimizers_mode", 0);
    if (want == 4) return run(3, R.maxlen, 0);
    const uint32_t smax = R.maxlen < MZ_SHORT_LEN ? R.maxlen : MZ_SHORT_LEN;
    if (R.minlen <= MZ_SHORT_LEN) {
        int mode = (KW == 1 && W <= 16) ? 2 : 0;
        if (want == 1) mode = 0; else if (want == 2 || ngsid_opt(ctx, "minimizers_lean", 0) != 0) mode = 1; else if (want == 3 && KW == 1 && W <= 32) mode = 2;
        if (mode == 0 && ((mz_lds_per_wave(smax, W, KW) + 15) & ~(size_t)15) * wpb > 160 * 1024) mode = 1;
        const int32_t rc = run(mode, smax, 0); if (rc) return rc;
    }
    if (R.maxlen > MZ_SHORT_LEN) { const int32_t rc = run(3, R.maxlen, MZ_SHORT_LEN); if (rc) return rc; }
    return NGSID_OK;
}

// Reads per launch: the kernel writes a read's minimizers at the read's base offset (no allocation pass, no atomics), i.e. into 12 (k > 21: 20) bytes per BASE.
// Round 5: that sparse image only ever exists for one chunk of at most MZ_CHUNK_BASES bases (3 GB of scratch); a gather per chunk appends the chunk to the compact
// CSR the consumers read (ctx->pol_mzcode / pol_mzpos through ctx->mz_off).  The counts come to the host chunk by chunk (the clustering driver needs them there
// anyway), the compact offsets are their running sum.
#define MZ_CHUNK_BASES ((uint64_t)256 << 20)

int32_t ngsid_minimizers_csr(ngsid_ctx* ctx, const DevReads& R, int k, int w, const MzOut& out, uint32_t* d_cnt, uint32_t* d_hlen, double* d_herr, double* d_rawerr,
                             uint32_t* h_cnt, uint32_t* h_hlen, long long* bad_read)
{
    DevBuf<uint64_t>& o_code = *out.code; DevBuf<uint32_t>& o_pos = *out.pos; DevBuf<uint64_t>& o_off = *out.off; PinVec<uint64_t>& o_hoff = *out.h_off;
    if (k < 1 || k > NGSID_MAX_K || w < k || w > 255) NGSID_FAIL(ctx, NGSID_ERR_ARG, "k must be in [1,%d] and k <= w <= 255 (k=%d w=%d)", NGSID_MAX_K, k, w);
    if (R.maxlen > NGSID_MAX_READ_LEN) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "read of %u bases exceeds NGSID_MAX_READ_LEN=%d", R.maxlen, NGSID_MAX_READ_LEN);
    if (!g_tables_loaded[ctx->device & 15]) {
        for (int qc = 0; qc < 127; ++qc)          // the run-quality choice of the kernel compares max(q, 33) instead of table entries: valid iff the table has this shape
            if (qc < 33 ? NGSID_PHRED_P[qc] != NGSID_PHRED_P[qc + 1] : !(NGSID_PHRED_P[qc] > NGSID_PHRED_P[qc + 1])) NGSID_FAIL(ctx, NGSID_ERR_ARG, "internal: phred table is not constant below '!' and strictly decreasing above");
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_phred_p), NGSID_PHRED_P, sizeof(double) * 128));
        g_tables_loaded[ctx->device & 15] = true;
    }
    if (bad_read) *bad_read = -1;
    const uint64_t n = R.n;
    o_hoff.resize(n + 1); o_hoff[0] = 0;
    HIPCHK(ctx, o_off.reserve(n + 1));
    if (n == 0) { HIPCHK(ctx, hipMemsetAsync(o_off.p, 0, 8, ctx->stream)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); return NGSID_OK; }
    const bool wide = k > 21;
    const long long chunk_opt = ngsid_opt(ctx, "minimizers_chunk_bases", 0);      // tests: many small chunks (results never depend on it)
    const uint64_t chunk_bases = chunk_opt > 0 ? (uint64_t)chunk_opt : std::max<uint64_t>(MZ_CHUNK_BASES / (uint64_t)ngsid_pool_contexts(), (uint64_t)32 << 20);      // (several contexts on one GPU: a share each)
    DevBuf<int> d_flag; HIPCHK(ctx, d_flag.alloc(1));
    DevBuf<uint64_t> s_hi, c_hi;                 // k > 21: second code word, sparse per chunk and compact over the call (until the rename pass)
    uint64_t done = 0, total = 0;                                  // reads finished, minimizers so far
    uint64_t cap_have = std::min<uint64_t>(o_code.p ? o_code.cap : 0, o_pos.p ? o_pos.cap : 0);
    if (wide) cap_have = 0;
    while (done < n) {
        uint64_t r1 = done + 1; const uint64_t b0 = R.h_off[done];
        while (r1 < n && R.h_off[r1 + 1] - b0 <= chunk_bases) ++r1;
        const uint64_t nr = r1 - done, cb = R.h_off[r1] - b0;
        HIPCHK(ctx, ctx->mz_scode.reserve(cb + 1)); HIPCHK(ctx, ctx->mz_spos.reserve(cb + 1)); if (wide) HIPCHK(ctx, s_hi.reserve(cb + 1));
        DevReads V; V.seq = R.seq; V.qual = R.qual; V.off = R.off + done; V.n = nr; V.total = cb; V.maxlen = R.maxlen; V.minlen = R.minlen;
        HIPCHK(ctx, hipMemsetAsync(d_flag.p, 0, sizeof(int), ctx->stream));
        int32_t rc = mz_launch(ctx, V, k, w, b0, ctx->mz_scode.p, wide ? s_hi.p : nullptr, ctx->mz_spos.p, d_cnt + done, d_hlen + done, d_herr + done, d_rawerr + done, d_flag.p);
        if (rc) return rc;
        int hf = 0;
        HIPCHK(ctx, hipMemcpyAsync(h_cnt + done, d_cnt + done, 4 * nr, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(h_hlen + done, d_hlen + done, 4 * nr, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(&hf, d_flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (hf && bad_read && *bad_read < 0) *bad_read = (long long)done + (hf - 1);
        uint64_t* ho = o_hoff.data();
        for (uint64_t r = done; r < r1; ++r) ho[r + 1] = ho[r] + h_cnt[r];
        total = ho[r1];
        if (total + 1 > cap_have) {         // grow the compact arrays, keeping what the earlier chunks wrote; sized for the rest of the call at this chunk's density (+ 12 %)
            const double dens = (double)(total - ho[done]) / (double)std::max<uint64_t>(cb, 1);
            const uint64_t want = total + (uint64_t)(dens * 1.125 * (double)(R.total - R.h_off[r1])) + 1024;
            o_code.n = o_pos.n = ho[done]; if (wide) c_hi.n = ho[done];
            HIPCHK(ctx, o_code.grow(want, ctx->stream)); HIPCHK(ctx, o_pos.grow(want, ctx->stream)); if (wide) HIPCHK(ctx, c_hi.grow(want, ctx->stream));
            cap_have = want;
        }
        HIPCHK(ctx, hipMemcpyAsync(o_off.p + done, ho + done, 8 * (nr + 1), hipMemcpyHostToDevice, ctx->stream));      // (pinned: h_mzoff stays valid and unchanged below r1)
        hipLaunchKernelGGL(k_mz_gather, dim3((unsigned)nr), dim3(64), 0, ctx->stream, R.off + done, d_cnt + done, o_off.p + done, nr, b0,
                           ctx->mz_scode.p, wide ? s_hi.p : nullptr, ctx->mz_spos.p, o_code.p, wide ? c_hi.p : nullptr, o_pos.p);
        HIPCHK(ctx, hipGetLastError());
        done = r1;
    }
    o_code.n = std::max<size_t>(o_code.n, total); o_pos.n = std::max<size_t>(o_pos.n, total);
    if (wide) return mz_rename_wide(ctx, o_code.p, c_hi.p, total);
    return NGSID_OK;
}

// ---- fingerprint of a device-resident read set (key of the context's minimizer cache, ngsid_internal.h: MzCache)
__global__ __launch_bounds__(256) void k_reads_fingerprint(const uint8_t* __restrict__ seq, uint64_t total, const uint64_t* __restrict__ off, uint64_t n, unsigned long long* __restrict__ out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x, nth = (uint64_t)gridDim.x * 256;
    unsigned long long h = 0;
    const uint64_t nw = total / 8;                                     // (hipMalloc'd / torch buffers are at least 8-byte aligned; a misaligned base takes the byte loop)
    if (((uintptr_t)seq & 7) == 0) {
        const unsigned long long* w8 = (const unsigned long long*)seq;
        for (uint64_t i = tid; i < nw; i += nth) { unsigned long long x = w8[i] + (i + 1) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; h += x; }
        for (uint64_t i = nw * 8 + tid; i < total; i += nth) h += ((unsigned long long)seq[i] + 1) * (i * 0x94D049BB133111EBull + 7);
    } else {
        for (uint64_t i = tid; i < total; i += nth) h += ((unsigned long long)seq[i] + 1) * (i * 0x94D049BB133111EBull + 7);
    }
    for (uint64_t i = tid; i <= n; i += nth) { unsigned long long x = off[i] + (i + 1) * 0xD6E8FEB86659FD93ull; x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; h += x; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) h += __shfl_xor(h, d);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

int32_t ngsid_reads_fingerprint(ngsid_ctx* ctx, const DevReads& R, unsigned long long* fp)
{
    if (ctx->mzc_fp.n < 1) HIPCHK(ctx, ctx->mzc_fp.alloc(1));
    HIPCHK(ctx, hipMemsetAsync(ctx->mzc_fp.p, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_reads_fingerprint, dim3(2048), dim3(256), 0, ctx->stream, R.seq, R.total, R.off, R.n, ctx->mzc_fp.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(fp, ctx->mzc_fp.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NGSID_OK;
}

