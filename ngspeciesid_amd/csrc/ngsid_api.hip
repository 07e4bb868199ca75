// ngsid_api.hip - C-ABI entry points that are thin (context, uploads, aligner batch, minimizer CSR, scoring)
#include "ngsid_internal.h"
#include "../../include/ngsid_merge_schedule.h"
#include "../../include/ngsid_tables.h"
#include <math.h>
#include <algorithm>

extern "C" uint32_t ngsid_abi_version(void) { return 2u; }

static char g_static_err[256] = "";
extern "C" const char* ngsid_last_error(ngsid_ctx* ctx) { return ctx ? ctx->err : g_static_err; }

// ---------------------------------------------------------------------------------------------- device memory cache
namespace {
struct DevPool { std::mutex mu; std::multimap<unsigned long long, void*> free_; size_t cached = 0; int contexts = 0; size_t live = 0, peak = 0;
                 struct Upload { size_t bs; void* q; size_t bq; void* off; size_t bo; }; std::map<void*, Upload> uploads;      // read sets handed out by ngsid_reads_upload
};
DevPool g_pool;
const size_t POOL_LIMIT = (size_t)48 << 30;          // bytes kept for reuse; beyond it blocks go back to the driver
const size_t POOL_HEADROOM = (size_t)3 << 30;        // device memory the library leaves to the runtime (ngsid_pool_alloc); at most a sixteenth of the device (small or partitioned GPUs)
// blocks taken from the pool by a function that can still fail: given back unless ownership is released (ngsid_reads_upload / _subset leaked them on a later error)
struct PoolBlocks { void* p[3] = {nullptr, nullptr, nullptr}; size_t b[3] = {0, 0, 0}; bool owned = true; ~PoolBlocks() { if (owned) for (int i = 0; i < 3; ++i) if (p[i]) ngsid_pool_free(p[i], b[i]); } };
inline size_t pool_class(size_t b) { if (b < 4096) return 4096; int sh = 63 - __builtin_clzll((unsigned long long)b) - 3; size_t m = ((size_t)1 << sh) - 1; return (b + m) & ~m; }
inline unsigned long long pool_key(int dev, size_t cls) { return ((unsigned long long)dev << 56) | (unsigned long long)cls; }
}
thread_local ngsid_ctx* g_ngsid_tls_ctx = nullptr;
hipError_t ngsid_pool_alloc(void** p, size_t bytes, size_t* got)
{
    const size_t cls = pool_class(bytes ? bytes : 1); int dev = 0; (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        auto it = g_pool.free_.find(pool_key(dev, cls));
        if (it != g_pool.free_.end()) { *p = it->second; g_pool.free_.erase(it); g_pool.cached -= cls; *got = cls; g_pool.live += cls; g_pool.peak = std::max(g_pool.peak, g_pool.live); return hipSuccess; }
    }
    // Headroom: the HIP / HSA runtime allocates device memory of its own while kernels run (scratch for register spills, queues, signals) and ABORTS the process
    // when it cannot ("HSA_STATUS_ERROR_OUT_OF_RESOURCES ... Available Free mem : 100 MB": eight contexts on one GPU, round 5).  The library therefore never takes the
    // last POOL_HEADROOM bytes: cached blocks go back to the driver first, then the request fails with hipErrorOutOfMemory - an error code instead of an abort.
    hipError_t e = hipErrorOutOfMemory;
    for (int attempt = 0; attempt < 2; ++attempt) {
        size_t freeb = 0, totalb = 0;
        const bool known = hipMemGetInfo(&freeb, &totalb) == hipSuccess;
        if (!known || freeb >= cls + std::min<size_t>(POOL_HEADROOM, totalb / 16)) { e = hipMalloc(p, cls); if (e == hipSuccess) break; (void)hipGetLastError(); }
        if (attempt == 0) ngsid_pool_release_all();          // out of memory (or too close to it): give the cached blocks back and retry once
    }
    *got = e == hipSuccess ? cls : 0;
    if (e == hipSuccess) { std::lock_guard<std::mutex> lk(g_pool.mu); g_pool.live += cls; g_pool.peak = std::max(g_pool.peak, g_pool.live); }
    return e;
}
void ngsid_pool_free(void* p, size_t bytes)
{
    if (!p) return;
    // like hipFree, giving a block back waits for the device: a buffer may be replaced (alloc / reserve / grow, error paths) while kernels that
    // use the old block are still in flight, and the block can be handed out again at once.  On an idle device this costs microseconds.
    // Round 6: inside an API call of a context (ApiClock) the wait is for that context's streams - nothing else can have used the block.
    if (ngsid_ctx* c = g_ngsid_tls_ctx) {
        (void)hipStreamSynchronize(c->stream);
        if (c->ev_fork) for (int i = 0; i < 4; ++i) if (c->side[i]) (void)hipStreamSynchronize(c->side[i]);
    } else (void)hipDeviceSynchronize();
    int dev = 0; (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> lk(g_pool.mu);
        g_pool.live -= std::min(g_pool.live, bytes);
        if (bytes && g_pool.cached + bytes <= POOL_LIMIT) { g_pool.free_.emplace(pool_key(dev, bytes), p); g_pool.cached += bytes; return; }
    }
    (void)hipFree(p);
}
// pinned host blocks of the PinVec staging vectors (ngsid_internal.h): same size classes as the device cache, freed blocks are kept (at most PIN_LIMIT bytes)
namespace { struct PinPool { std::mutex mu; std::multimap<size_t, void*> free_; size_t cached = 0; }; PinPool g_pin; const size_t PIN_LIMIT = (size_t)4 << 30; }
void* ngsid_pinned_alloc(size_t bytes)
{
    const size_t cls = pool_class(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(g_pin.mu);
        auto it = g_pin.free_.find(cls);
        if (it != g_pin.free_.end()) { void* p = it->second; g_pin.free_.erase(it); g_pin.cached -= cls; return p; }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, cls, hipHostMallocDefault) != hipSuccess) {          // give the cached blocks back and try once more
        (void)hipGetLastError();
        std::vector<void*> old; { std::lock_guard<std::mutex> lk(g_pin.mu); for (auto& kv : g_pin.free_) old.push_back(kv.second); g_pin.free_.clear(); g_pin.cached = 0; }
        for (void* q : old) (void)hipHostFree(q);
        if (hipHostMalloc(&p, cls, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    return p;
}
void ngsid_pinned_free(void* p, size_t bytes)
{
    if (!p) return;
    const size_t cls = pool_class(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(g_pin.mu);
        if (g_pin.cached + cls <= PIN_LIMIT) { g_pin.free_.emplace(cls, p); g_pin.cached += cls; return; }
    }
    (void)hipHostFree(p);
}
size_t ngsid_pool_cached_bytes() { std::lock_guard<std::mutex> lk(g_pool.mu); return g_pool.cached; }
int ngsid_pool_contexts() { std::lock_guard<std::mutex> lk(g_pool.mu); return g_pool.contexts > 0 ? g_pool.contexts : 1; }
void ngsid_pool_stats(size_t* live, size_t* peak, bool reset_peak) { std::lock_guard<std::mutex> lk(g_pool.mu); if (live) *live = g_pool.live; if (peak) *peak = g_pool.peak; if (reset_peak) g_pool.peak = g_pool.live; }
void ngsid_pool_release_all()
{
    std::lock_guard<std::mutex> lk(g_pool.mu);
    for (auto& kv : g_pool.free_) (void)hipFree(kv.second);
    g_pool.free_.clear(); g_pool.cached = 0;
}

extern "C" int32_t ngsid_create(int32_t device_ordinal, uint32_t flags, ngsid_ctx** out)
{
    (void)flags;
    if (!out) return NGSID_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { snprintf(g_static_err, sizeof g_static_err, "no HIP device (%s); this library has no CPU path", hipGetErrorString(e)); return NGSID_ERR_NO_DEVICE; }
    if (device_ordinal < 0 || device_ordinal >= ndev) { snprintf(g_static_err, sizeof g_static_err, "device ordinal %d out of range (0..%d)", device_ordinal, ndev - 1); return NGSID_ERR_ARG; }
    e = hipSetDevice(device_ordinal);
    if (e != hipSuccess) { snprintf(g_static_err, sizeof g_static_err, "hipSetDevice: %s", hipGetErrorString(e)); return NGSID_ERR_HIP; }
    ngsid_ctx* c = new ngsid_ctx();
    c->device = device_ordinal;
    c->debug_sync = getenv("NGSID_DEBUG_SYNC") != nullptr;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) {
        c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        size_t freeb = 0, totalb = 0;
        if (hipMemGetInfo(&freeb, &totalb) == hipSuccess && freeb > ((size_t)8 << 30)) c->scratch_budget = std::min<size_t>(freeb / 4, (size_t)32 << 30);
    }
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { snprintf(g_static_err, sizeof g_static_err, "hipStreamCreate: %s", hipGetErrorString(e)); delete c; return NGSID_ERR_HIP; }
    { std::lock_guard<std::mutex> lk(g_pool.mu); g_pool.contexts++; }
    *out = c;
    return NGSID_OK;
}

extern "C" void ngsid_destroy(ngsid_ctx* ctx)
{
    if (!ctx) return;
    if (g_ngsid_tls_ctx == ctx) g_ngsid_tls_ctx = nullptr;
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < 4; ++i) { if (ctx->side[i]) { (void)hipStreamSynchronize(ctx->side[i]); (void)hipStreamDestroy(ctx->side[i]); } if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]); }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
    bool last; { std::lock_guard<std::mutex> lk(g_pool.mu); last = --g_pool.contexts <= 0; }
    if (last) {
        {   // read sets nobody released: their buffers go with the last context
            std::lock_guard<std::mutex> lk(g_pool.mu);
            for (auto& kv : g_pool.uploads) { (void)hipFree(kv.first); if (kv.second.q) (void)hipFree(kv.second.q); (void)hipFree(kv.second.off); }
            g_pool.uploads.clear();
        }
        ngsid_pool_release_all();
        std::vector<void*> pins; { std::lock_guard<std::mutex> lk(g_pin.mu); for (auto& kv : g_pin.free_) pins.push_back(kv.second); g_pin.free_.clear(); g_pin.cached = 0; }
        for (void* q : pins) (void)hipHostFree(q);          // (the pinned staging blocks no vector holds any more)
    }
}

int32_t ngsid_side_streams(ngsid_ctx* ctx)
{
    if (ctx->ev_fork) return NGSID_OK;
    for (int i = 0; i < 4; ++i) { HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking)); HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming)); }
    HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    return NGSID_OK;
}

__global__ __launch_bounds__(256) void k_off_fingerprint(const uint64_t* __restrict__ off, uint64_t n, unsigned long long* __restrict__ out)
{
    unsigned long long h = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * 256) { unsigned long long x = off[i] + (i + 1) * 0xD6E8FEB86659FD93ull; x ^= x >> 31; x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29; h += x; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) h += __shfl_xor(h, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

int32_t ngsid_upload_reads(ngsid_ctx* ctx, const ngsid_reads_t* in, DevReads* out, bool need_qual)
{
    if (!in || (!in->off) || (in->n && !in->seq)) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null read set");
    if (need_qual && in->n && !in->qual) NGSID_FAIL(ctx, NGSID_ERR_ARG, "this entry point needs qualities");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HostTimer hu(ctx->stream, "upload_reads");
    out->n = in->n;
    if (in->mem == NGSID_MEM_DEVICE) {
        out->seq = in->seq; out->qual = in->qual; out->off = in->off;
        // Round 5: the context remembers the host copy of the offsets of the last device read set (same pointer, same count, same 64-bit fingerprint computed on the
        // device): the three calls of a pass (cluster, draft, polish) download and scan 8 MB per million reads ONCE.  (The first large device -> host copy after the few
        // idle milliseconds between the clustering and the consensus call was also seen to take 10 - 30 ms instead of 0.16 ms on most runs: profiles/NOTES.md.)
        unsigned long long fp = 0;
        {
            if (ctx->mzc_fp.n < 2) HIPCHK(ctx, ctx->mzc_fp.alloc(2));
            HIPCHK(ctx, hipMemsetAsync(ctx->mzc_fp.p + 1, 0, sizeof(unsigned long long), ctx->stream));
            hipLaunchKernelGGL(k_off_fingerprint, dim3(256), dim3(256), 0, ctx->stream, in->off, in->n, ctx->mzc_fp.p + 1);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemcpyAsync(&fp, ctx->mzc_fp.p + 1, sizeof fp, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
        if (ctx->offc.v && ctx->offc.ptr == (const void*)in->off && ctx->offc.n == in->n && ctx->offc.fp == fp && ctx->offc.v->size() == in->n + 1) {
            out->h_off.v = ctx->offc.v; out->total = (*out->h_off.v)[in->n]; out->maxlen = ctx->offc.maxlen; out->minlen = ctx->offc.minlen;
            hu.mark("offsets: cached");
            return NGSID_OK;
        }
        out->h_off.resize(in->n + 1);
        hu.mark("resize");
        // through pinned staging on the context's stream: a blocking copy into fresh pageable memory cost 26 ms for 8 MB here
        const size_t nb = sizeof(uint64_t) * (in->n + 1);
        if (ctx->pin_bytes < nb) { if (ctx->pin) (void)hipHostFree(ctx->pin); ctx->pin = nullptr; ctx->pin_bytes = 0; HIPCHK(ctx, hipHostMalloc(&ctx->pin, nb + nb / 8, hipHostMallocDefault)); ctx->pin_bytes = nb + nb / 8; }
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin, in->off, nb, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        hu.mark("d2h");
        memcpy(out->h_off.data(), ctx->pin, nb);
        hu.mark("memcpy");
        ctx->offc.ptr = in->off; ctx->offc.n = in->n; ctx->offc.fp = fp; ctx->offc.v = out->h_off.v; ctx->offc.maxlen = 0;       // (lengths filled in below, after the scan)
    } else {
        out->h_off.resize(in->n + 1);
        memcpy(out->h_off.data(), in->off, sizeof(uint64_t) * (in->n + 1));
        const uint64_t total = out->h_off[in->n];
        HIPCHK(ctx, out->own_seq.alloc(total + 16)); HIPCHK(ctx, out->own_off.alloc(in->n + 1));
        if (total) HIPCHK(ctx, hipMemcpyAsync(out->own_seq.p, in->seq, total, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(out->own_off.p, in->off, sizeof(uint64_t) * (in->n + 1), hipMemcpyHostToDevice, ctx->stream));
        out->seq = out->own_seq.p; out->off = out->own_off.p; out->qual = nullptr;
        if (in->qual) { HIPCHK(ctx, out->own_qual.alloc(total + 16)); if (total) HIPCHK(ctx, hipMemcpyAsync(out->own_qual.p, in->qual, total, hipMemcpyHostToDevice, ctx->stream)); out->qual = out->own_qual.p; }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    out->total = out->h_off[in->n];
    uint32_t mx = 0, mn = 0xffffffffu;
    for (uint64_t i = 0; i < in->n; ++i) {
        if (out->h_off[i + 1] < out->h_off[i]) NGSID_FAIL(ctx, NGSID_ERR_ARG, "offsets not monotone at read %llu", (unsigned long long)i);
        const uint64_t l = out->h_off[i + 1] - out->h_off[i];
        if (l > 0xffffffffull) NGSID_FAIL(ctx, NGSID_ERR_TOO_LONG, "read %llu too long", (unsigned long long)i);
        mx = std::max<uint32_t>(mx, (uint32_t)l); mn = std::min<uint32_t>(mn, (uint32_t)l);
    }
    out->maxlen = mx; out->minlen = in->n ? mn : 0;
    if (in->mem == NGSID_MEM_DEVICE && ctx->offc.v == out->h_off.v) { ctx->offc.maxlen = out->maxlen; ctx->offc.minlen = out->minlen; }
    hu.mark("scan");
    return NGSID_OK;
}

// ---------------------------------------------------------------------------------------------- (a10,a15)
// (a10, section 8b) the alignment itself: what parasail returns as result.cigar (cluster.py:138-144, consensus.py:64-73)
extern "C" int32_t ngsid_sg_align_cigar_batch(ngsid_ctx* ctx, const ngsid_reads_t* queries, const ngsid_reads_t* targets,
                                              const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                                              int32_t match, int32_t mismatch, const int32_t* open, int32_t ext,
                                              int32_t* score, uint64_t* ops_off, uint8_t* ops, uint64_t cap, uint64_t* needed)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!queries || !targets || !ops_off || (n_pairs && (!q_idx || !t_idx || !open))) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    DevReads Q, T;
    int32_t rc = ngsid_upload_reads(ctx, queries, &Q, false); if (rc) return rc;
    rc = ngsid_upload_reads(ctx, targets, &T, false); if (rc) return rc;
    ops_off[0] = 0;
    if (n_pairs == 0) { if (needed) *needed = 0; return NGSID_OK; }
    uint32_t mq = 0, mt = 0;
    std::vector<uint64_t> h_off(n_pairs + 1, 0);
    for (uint64_t p = 0; p < n_pairs; ++p) {
        if (q_idx[p] >= Q.n || t_idx[p] >= T.n) NGSID_FAIL(ctx, NGSID_ERR_ARG, "pair %llu out of range", (unsigned long long)p);
        const uint32_t ql = (uint32_t)(Q.h_off[q_idx[p] + 1] - Q.h_off[q_idx[p]]), tl = (uint32_t)(T.h_off[t_idx[p] + 1] - T.h_off[t_idx[p]]);
        mq = std::max(mq, ql); mt = std::max(mt, tl);
        h_off[p + 1] = h_off[p] + ql + tl;                    // capacity of the pair: every column consumes at least one base
    }
    DevBuf<uint32_t> dq, dt; DevBuf<int32_t> dopen, dout; DevBuf<uint64_t> doff; DevBuf<uint8_t> dops;
    HIPCHK(ctx, dq.alloc(n_pairs)); HIPCHK(ctx, dt.alloc(n_pairs)); HIPCHK(ctx, dopen.alloc(n_pairs)); HIPCHK(ctx, dout.alloc(n_pairs * 4)); HIPCHK(ctx, doff.alloc(n_pairs + 1)); HIPCHK(ctx, dops.alloc(h_off[n_pairs] + 1));
    HIPCHK(ctx, hipMemcpyAsync(dq.p, q_idx, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dt.p, t_idx, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dopen.p, open, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(doff.p, h_off.data(), 8 * (n_pairs + 1), hipMemcpyHostToDevice, ctx->stream));
    AlignJob J{};
    J.qseq = Q.seq; J.qoff = Q.off; J.tseq = T.seq; J.toff = T.off; J.qidx = dq.p; J.tidx = dt.p; J.npairs = n_pairs;
    J.match = match; J.mismatch = mismatch; J.ext = ext; J.k = 1; J.open = dopen.p; J.match_id = nullptr;
    J.score = dout.p; J.ncols = dout.p + n_pairs; J.nmatch = nullptr; J.region = nullptr; J.bp = nullptr; J.bp_windows = 0; J.window = 1; J.span = nullptr;
    J.ops = dops.p; J.ops_off = doff.p;
    int mo = 0; for (uint64_t p = 0; p < n_pairs; ++p) { if (open[p] < 0) { mo = 1 << 20; break; } mo = std::max(mo, (int)open[p]); }
    rc = ngsid_launch_align(ctx, J, mq, mt, mo); if (rc) return rc;
    std::vector<int32_t> h(n_pairs * 2); std::vector<uint8_t> hops(h_off[n_pairs] + 1);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), dout.p, 8 * n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(hops.data(), dops.p, h_off[n_pairs], hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (score) memcpy(score, h.data(), 4 * n_pairs);
    uint64_t total = 0; for (uint64_t p = 0; p < n_pairs; ++p) { total += (uint64_t)h[n_pairs + p]; ops_off[p + 1] = total; }
    if (needed) *needed = total;
    if (total > cap || (!ops && total)) NGSID_FAIL(ctx, NGSID_ERR_CAPACITY, "ops buffer too small: need %llu bytes", (unsigned long long)total);
    static const uint8_t sym[4] = {'=', 'X', 'I', 'D'};
    for (uint64_t p = 0; p < n_pairs; ++p) {            // the kernel wrote the columns in traceback order: reverse them into alignment order
        const uint64_t c = (uint64_t)h[n_pairs + p]; const uint8_t* src = hops.data() + h_off[p]; uint8_t* dst = ops + ops_off[p];
        const uint64_t ql = Q.h_off[q_idx[p] + 1] - Q.h_off[q_idx[p]], tl = T.h_off[t_idx[p] + 1] - T.h_off[t_idx[p]];
        if (ql == 0 || tl == 0) { for (uint64_t x = 0; x < c; ++x) dst[x] = ql ? 'I' : 'D'; continue; }      // an empty sequence: the kernel has no cell to trace, the other one is all end gap
        for (uint64_t x = 0; x < c; ++x) dst[x] = sym[src[c - 1 - x] & 3];
    }
    return NGSID_OK;
}

extern "C" int32_t ngsid_sg_align_batch(ngsid_ctx* ctx, const ngsid_reads_t* queries, const ngsid_reads_t* targets,
                                        const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                                        int32_t match, int32_t mismatch, const int32_t* open, int32_t ext,
                                        int32_t k, const int32_t* match_id,
                                        int32_t* score, int32_t* n_cols, int32_t* n_match, int32_t* region)
{
    ApiClock api_clock_(ctx, "sg_align_batch");
    if (!ctx) return NGSID_ERR_ARG;
    if (!queries || !targets || (n_pairs && (!q_idx || !t_idx || !open))) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    DevReads Q, T;
    int32_t rc = ngsid_upload_reads(ctx, queries, &Q, false); if (rc) return rc;
    rc = ngsid_upload_reads(ctx, targets, &T, false); if (rc) return rc;
    if (n_pairs == 0) return NGSID_OK;
    uint32_t mq = 0, mt = 0;
    for (uint64_t p = 0; p < n_pairs; ++p) {
        if (q_idx[p] >= Q.n || t_idx[p] >= T.n) NGSID_FAIL(ctx, NGSID_ERR_ARG, "pair %llu out of range", (unsigned long long)p);
        mq = std::max<uint32_t>(mq, (uint32_t)(Q.h_off[q_idx[p] + 1] - Q.h_off[q_idx[p]]));
        mt = std::max<uint32_t>(mt, (uint32_t)(T.h_off[t_idx[p] + 1] - T.h_off[t_idx[p]]));
    }
    DevBuf<uint32_t> dq, dt; DevBuf<int32_t> dopen, dmid, dout;
    HIPCHK(ctx, dq.alloc(n_pairs)); HIPCHK(ctx, dt.alloc(n_pairs)); HIPCHK(ctx, dopen.alloc(n_pairs)); HIPCHK(ctx, dmid.alloc(n_pairs)); HIPCHK(ctx, dout.alloc(n_pairs * 4));
    HIPCHK(ctx, hipMemcpyAsync(dq.p, q_idx, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dt.p, t_idx, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dopen.p, open, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    if (match_id) HIPCHK(ctx, hipMemcpyAsync(dmid.p, match_id, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    AlignJob J{};
    J.qseq = Q.seq; J.qoff = Q.off; J.tseq = T.seq; J.toff = T.off; J.qidx = dq.p; J.tidx = dt.p; J.npairs = n_pairs;
    J.match = match; J.mismatch = mismatch; J.ext = ext; J.k = k; J.open = dopen.p; J.match_id = match_id ? dmid.p : nullptr;
    J.score = dout.p; J.ncols = dout.p + n_pairs; J.nmatch = dout.p + 2 * n_pairs; J.region = dout.p + 3 * n_pairs;
    J.bp = nullptr; J.bp_windows = 0; J.window = 1; J.span = nullptr;
    int mo = 0; for (uint64_t p = 0; p < n_pairs; ++p) { if (open[p] < 0) { mo = 1 << 20; break; } mo = std::max(mo, (int)open[p]); }
    rc = ngsid_launch_align(ctx, J, mq, mt, mo); if (rc) return rc;
    std::vector<int32_t> h(n_pairs * 4);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), dout.p, 16 * n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (score) memcpy(score, h.data(), 4 * n_pairs);
    if (n_cols) memcpy(n_cols, h.data() + n_pairs, 4 * n_pairs);
    if (n_match) memcpy(n_match, h.data() + 2 * n_pairs, 4 * n_pairs);
    if (region) memcpy(region, h.data() + 3 * n_pairs, 4 * n_pairs);
    return NGSID_OK;
}

// ---------------------------------------------------------------------------------------------- (a17, edit-distance mode)
extern "C" int32_t ngsid_ed_align_batch(ngsid_ctx* ctx, const ngsid_reads_t* queries, const ngsid_reads_t* targets,
                                        const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                                        int32_t window, int32_t bp_windows, int32_t* distance, int32_t* span, int32_t* bp)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!queries || !targets || (n_pairs && (!q_idx || !t_idx))) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    if (bp && (bp_windows <= 0 || window <= 0)) NGSID_FAIL(ctx, NGSID_ERR_ARG, "bp needs window > 0 and bp_windows > 0");
    DevReads Q, T;
    int32_t rc = ngsid_upload_reads(ctx, queries, &Q, false); if (rc) return rc;
    rc = ngsid_upload_reads(ctx, targets, &T, false); if (rc) return rc;
    if (n_pairs == 0) return NGSID_OK;
    uint32_t mq = 0, mt = 0;
    for (uint64_t p = 0; p < n_pairs; ++p) {
        if (q_idx[p] >= Q.n || t_idx[p] >= T.n) NGSID_FAIL(ctx, NGSID_ERR_ARG, "pair %llu out of range", (unsigned long long)p);
        mq = std::max<uint32_t>(mq, (uint32_t)(Q.h_off[q_idx[p] + 1] - Q.h_off[q_idx[p]]));
        mt = std::max<uint32_t>(mt, (uint32_t)(T.h_off[t_idx[p] + 1] - T.h_off[t_idx[p]]));
    }
    DevBuf<uint32_t> dq, dt; DevBuf<int32_t> ddist, dspan, dbp;
    HIPCHK(ctx, dq.alloc(n_pairs)); HIPCHK(ctx, dt.alloc(n_pairs)); HIPCHK(ctx, ddist.alloc(n_pairs)); HIPCHK(ctx, dspan.alloc(n_pairs * 4));
    if (bp) HIPCHK(ctx, dbp.alloc(n_pairs * (uint64_t)bp_windows * 4));
    HIPCHK(ctx, hipMemcpyAsync(dq.p, q_idx, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dt.p, t_idx, 4 * n_pairs, hipMemcpyHostToDevice, ctx->stream));
    AlignJob J{};
    J.qseq = Q.seq; J.qoff = Q.off; J.tseq = T.seq; J.toff = T.off; J.qidx = dq.p; J.tidx = dt.p; J.npairs = n_pairs;
    J.bp = bp ? dbp.p : nullptr; J.bp_windows = bp ? bp_windows : 0; J.window = window; J.span = dspan.p;
    rc = ngsid_launch_ed_align(ctx, J, mq, mt, ddist.p); if (rc) return rc;
    if (distance) HIPCHK(ctx, hipMemcpyAsync(distance, ddist.p, 4 * n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    if (span) HIPCHK(ctx, hipMemcpyAsync(span, dspan.p, 16 * n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    if (bp) HIPCHK(ctx, hipMemcpyAsync(bp, dbp.p, 16ull * n_pairs * bp_windows, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NGSID_OK;
}

// ---------------------------------------------------------------------------------------------- (a1-a3)
extern "C" int32_t ngsid_hpc_minimizers(ngsid_ctx* ctx, const ngsid_reads_t* reads, int32_t k, int32_t w,
                                        uint64_t* mz_off, uint64_t* codes, uint32_t* pos, uint64_t cap, uint64_t* needed,
                                        uint32_t* hpc_len, double* hpc_err)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!reads || !mz_off) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    DevReads R; int32_t rc = ngsid_upload_reads(ctx, reads, &R, false); if (rc) return rc;
    const uint64_t n = R.n;
    DevBuf<uint64_t> ccode, coff; DevBuf<uint32_t> cpos, dcnt, dhl; DevBuf<double> dherr, draw; PinVec<uint64_t> hmoff; PinVec<uint32_t> hcnt(n), hhl(n);
    HIPCHK(ctx, dcnt.alloc(n)); HIPCHK(ctx, dhl.alloc(n)); HIPCHK(ctx, dherr.alloc(n)); HIPCHK(ctx, draw.alloc(n));
    long long bad = -1;
    rc = ngsid_minimizers_csr(ctx, R, k, w, MzOut{&ccode, &cpos, &coff, &hmoff}, dcnt.p, dhl.p, dherr.p, draw.p, hcnt.data(), hhl.data(), &bad); if (rc) return rc;      // (chunked: the sparse image of the kernel never exceeds 3 GB)
    if (bad >= 0) NGSID_FAIL(ctx, NGSID_ERR_ALPHABET, "read %lld: base outside ACGTN", bad);
    std::vector<double> hherr(n);
    if (n) HIPCHK(ctx, hipMemcpyAsync(hherr.data(), dherr.p, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t total = hmoff[n];
    if (needed) *needed = total;
    if (hpc_len) memcpy(hpc_len, hhl.data(), 4 * n);
    if (hpc_err) memcpy(hpc_err, hherr.data(), 8 * n);
    const bool dev_out = reads->mem == NGSID_MEM_DEVICE;
    if (total > cap || (total && (!codes || !pos))) {
        if (!dev_out) memcpy(mz_off, hmoff.data(), 8 * (n + 1));
        NGSID_FAIL(ctx, NGSID_ERR_CAPACITY, "minimizer buffer too small: need %llu entries", (unsigned long long)total);
    }
    const hipMemcpyKind kind = dev_out ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (dev_out) HIPCHK(ctx, hipMemcpyAsync(mz_off, coff.p, 8 * (n + 1), kind, ctx->stream)); else memcpy(mz_off, hmoff.data(), 8 * (n + 1));
    if (total) { HIPCHK(ctx, hipMemcpyAsync(codes, ccode.p, 8 * total, kind, ctx->stream)); HIPCHK(ctx, hipMemcpyAsync(pos, cpos.p, 4 * total, kind, ctx->stream)); }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NGSID_OK;
}

// ---------------------------------------------------------------------------------------------- (f1)
// get_sorted_fastq_for_cluster.py:23-33,124-155.  One LANE per read (64 reads per wave): the sliding product
// cur *= (1-p_new)/(1-p_old) is a sequential recurrence that must be replayed op for op to give the reference's exact double; reads are
// independent, so the batch is the parallel axis.  The bases and qualities of the 64 reads pass through LDS in 64-byte slices (dword loads,
// 16 lanes per read and instruction, instead of 64 scattered byte loads), the per-read quality histogram lives in LDS ([bin][lane]) and the
// two probability tables are read from LDS as well.
__constant__ double c_p_clamped[128];
__constant__ double c_p_nomin[128];
static bool g_score_tables[16] = {false};

#define SC_QS 132      /* quality ring row: two 64-byte slices + pad (33 dwords: lanes hit different banks) */
#define SC_SS 68       /* sequence slice row */
template <typename HT>       // histogram counter: 16 bits for reads up to 65 535 bases, 32 bits beyond (round 5: such reads are scored and written, not clustered)
__global__ __launch_bounds__(64)
void k_score_reads(const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, const uint64_t* __restrict__ off, uint64_t n,
                   int k, double qthr, double* __restrict__ score, double* __restrict__ err, uint8_t* __restrict__ keep)
{
    __shared__ __attribute__((aligned(16))) uint8_t qt[64 * SC_QS];
    __shared__ __attribute__((aligned(16))) uint8_t st[64 * SC_SS];
    __shared__ HT hist[128 * 64];
    __shared__ double tp[128], tn[128];
    const int lane = threadIdx.x;
    const uint64_t r = (uint64_t)blockIdx.x * 64 + lane;
    const bool have = r < n;
    const uint64_t b = have ? off[r] : 0; const int len = have ? (int)(off[r + 1] - b) : 0;
    tp[lane] = c_p_clamped[lane]; tp[lane + 64] = c_p_clamped[lane + 64]; tn[lane] = c_p_nomin[lane]; tn[lane + 64] = c_p_nomin[lane + 64];
    for (int c = 0; c < 128; ++c) hist[c * 64 + lane] = 0;
    int maxlen = len;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, d));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    const bool active = have && len >= 2 * k;               // :142 (reads shorter than 2k are dropped before anything is computed)
    int hl = 0; uint8_t prevc = 0;
    double cur = 1.0, sum = 0.0;
    const int sub = lane >> 4, dj = lane & 15;               // slice loads: 4 reads per instruction, 16 dwords each
    for (int c0 = 0; c0 < maxlen; c0 += 64) {
        const int half = c0 & 64;
        for (int it = 0; it < 16; ++it) {
            const int rr = it * 4 + sub;
            const uint64_t rb = __shfl((unsigned long long)b, rr); const int rl = __shfl(len, rr);
            const int x = c0 + dj * 4;
            unsigned qw = 0, sw = 0;
            if (x + 3 < rl) { qw = *(const unsigned*)(qual + rb + x); sw = *(const unsigned*)(seq + rb + x); }
            else for (int y = 0; y < 4; ++y) if (x + y < rl) { qw |= (unsigned)qual[rb + x + y] << (8 * y); sw |= (unsigned)seq[rb + x + y] << (8 * y); }
            *(unsigned*)(qt + rr * SC_QS + half + dj * 4) = qw;
            *(unsigned*)(st + rr * SC_SS + dj * 4) = sw;
        }
        __syncthreads();
        if (active) {
            const int e = min(64, len - c0);
            for (int i = 0; i < e; ++i) {
                const int gi = c0 + i;
                const uint8_t sc = st[lane * SC_SS + i], qc = qt[lane * SC_QS + half + i] & 127;
                hl += (gi == 0 || sc != prevc); prevc = sc;
                hist[qc * 64 + lane] += 1;
                const double pn = 1.0 - tp[qc];
                if (gi < k) { cur = cur * pn; if (gi == k - 1) sum = cur; }
                else { const double leave = 1.0 - tp[qt[lane * SC_QS + ((gi - k) & 127)] & 127]; cur *= (pn / leave); sum += cur; }
            }
        }
        __syncthreads();
    }
    if (!have) return;
    score[r] = 0.0; err[r] = 0.0; keep[r] = 0;
    if (!active || hl < k) return;
    const double ee = (double)(len - k + 1) - sum;
    const double pno = 1.0 - ee / (double)(len - k + 1);
    score[r] = pno * (double)(len - k + 1);
    double se = 0.0;
    for (int c = 0; c < 128; ++c) { const int h = hist[c * 64 + lane]; if (h) se = se + (double)h * tn[c]; }
    const double er = se / (double)len;
    err[r] = er;
    if (10.0 * -(log(er) / log(10.0)) <= qthr) return;      // device log(): last-ulp differences only matter exactly on the threshold
    keep[r] = 1;
}

extern "C" int32_t ngsid_score_reads(ngsid_ctx* ctx, const ngsid_reads_t* reads, int32_t k, double q_threshold,
                                     double* score, double* err_rate, uint8_t* keep)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!reads || !score || !err_rate || !keep || k < 1) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    if (k > 64) NGSID_FAIL(ctx, NGSID_ERR_ARG, "score_reads: k = %d exceeds 64 (the quality ring of the kernel)", (int)k);
    DevReads R; int32_t rc = ngsid_upload_reads(ctx, reads, &R, true); if (rc) return rc;
    if (!g_score_tables[ctx->device & 15]) {
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_p_clamped), NGSID_PHRED_P, sizeof(double) * 128));
        HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_p_nomin), NGSID_PHRED_P_NOMIN, sizeof(double) * 128));
        g_score_tables[ctx->device & 15] = true;
    }
    const uint64_t n = R.n; if (!n) return NGSID_OK;
    DevBuf<double> ds, de; DevBuf<uint8_t> dk;
    HIPCHK(ctx, ds.alloc(n)); HIPCHK(ctx, de.alloc(n)); HIPCHK(ctx, dk.alloc(n));
    { ProfScope ps_(ctx, "k_score_reads");
      if (R.maxlen <= 65535u) hipLaunchKernelGGL(k_score_reads<unsigned short>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, R.seq, R.qual, R.off, n, k, q_threshold, ds.p, de.p, dk.p);
      else hipLaunchKernelGGL(k_score_reads<unsigned int>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, R.seq, R.qual, R.off, n, k, q_threshold, ds.p, de.p, dk.p); }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(score, ds.p, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(err_rate, de.p, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(keep, dk.p, n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return NGSID_OK;
}

// ---------------------------------------------------------------------------------------------- measurement hooks
static void prof_collect(ngsid_ctx* ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& e : ctx->prof_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { auto& acc = ctx->prof_acc[e.name]; acc.first += ms; acc.second += 1; }
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    ctx->prof_events.clear();
}
extern "C" int32_t ngsid_profile_enable(ngsid_ctx* ctx, int32_t on)
{
    if (!ctx) return NGSID_ERR_ARG;
    prof_collect(ctx); ctx->prof_acc.clear(); ctx->prof = on != 0;
    if (on) { if (ctx->stat.n < 8) HIPCHK(ctx, ctx->stat.alloc(8)); HIPCHK(ctx, hipMemsetAsync(ctx->stat.p, 0, 8 * sizeof(unsigned long long), ctx->stream)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }
    return NGSID_OK;
}
extern "C" int32_t ngsid_reads_upload(ngsid_ctx* ctx, const ngsid_reads_t* host, ngsid_reads_t* dev)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!host || !dev || !host->off || (host->n && !host->seq) || host->mem != NGSID_MEM_HOST) NGSID_FAIL(ctx, NGSID_ERR_ARG, "reads_upload: a host read set and an output are required");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = host->n, total = host->off[n];
    PoolBlocks PB; void *&ds = PB.p[0], *&dq = PB.p[1], *&dof = PB.p[2]; size_t &bs = PB.b[0], &bq = PB.b[1], &bo = PB.b[2];
    HIPCHK(ctx, ngsid_pool_alloc(&ds, total + 16, &bs));
    HIPCHK(ctx, ngsid_pool_alloc(&dof, sizeof(uint64_t) * (n + 1), &bo));
    if (host->qual) HIPCHK(ctx, ngsid_pool_alloc(&dq, total + 16, &bq));
    if (total) HIPCHK(ctx, hipMemcpyAsync(ds, host->seq, total, hipMemcpyHostToDevice, ctx->stream));
    if (total && host->qual) HIPCHK(ctx, hipMemcpyAsync(dq, host->qual, total, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dof, host->off, sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    { std::lock_guard<std::mutex> lk(g_pool.mu); g_pool.uploads[ds] = {bs, dq, bq, dof, bo}; }
    PB.owned = false;
    dev->seq = (const uint8_t*)ds; dev->qual = (const uint8_t*)dq; dev->off = (const uint64_t*)dof; dev->n = n; dev->mem = NGSID_MEM_DEVICE; dev->_pad = 0;
    return NGSID_OK;
}


// (b, f2) the reads idx[0..n) of a DEVICE-resident read set, in that order, as a new device-resident read set (the CLI: the reads cross PCIe once in FILE order,
// the score order is a gather on the device - get_sorted_fastq_for_cluster.py:174 sorts in host memory).  *foreign (may be NULL) = bases of the result outside
// A/C/G/T/N (what ngsid_host_count_foreign_bases counts: the caller normalises a copy only when there are any).
__global__ __launch_bounds__(256) void k_reads_subset(const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, const uint64_t* __restrict__ off, const uint64_t* __restrict__ idx,
                                                      const uint64_t* __restrict__ noff, uint64_t n, uint8_t* __restrict__ oseq, uint8_t* __restrict__ oqual, unsigned long long* __restrict__ foreign)
{
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const uint64_t src = off[idx[r]], dst = noff[r]; const uint64_t len = noff[r + 1] - dst;
    unsigned bad = 0;
    for (uint64_t x = lane; x < len; x += 64) {
        const uint8_t c = seq[src + x]; oseq[dst + x] = c; if (oqual) oqual[dst + x] = qual[src + x];
        bad += !(c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N');
    }
    if (foreign) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) bad += __shfl_xor(bad, d);
        if (lane == 0 && bad) atomicAdd(foreign, (unsigned long long)bad);
    }
}
extern "C" int32_t ngsid_reads_subset(ngsid_ctx* ctx, const ngsid_reads_t* dev_in, const uint64_t* idx, uint64_t n, ngsid_reads_t* dev_out, uint64_t* foreign)
{
    if (!ctx) return NGSID_ERR_ARG;
    ApiClock api_clock_(ctx, "reads_subset");          // (also: the temporaries freed below wait for THIS context's streams, not for the device - two contexts cluster side by side)
    if (!dev_in || !dev_out || (n && !idx) || dev_in->mem != NGSID_MEM_DEVICE) NGSID_FAIL(ctx, NGSID_ERR_ARG, "reads_subset: a device read set, an index list and an output are required");
    DevReads R; int32_t rc = ngsid_upload_reads(ctx, dev_in, &R, false); if (rc) return rc;
    static thread_local PinVec<uint64_t> h_noff; h_noff.resize(n + 1); h_noff[0] = 0;
    for (uint64_t i = 0; i < n; ++i) { if (idx[i] >= R.n) NGSID_FAIL(ctx, NGSID_ERR_ARG, "reads_subset: index %llu out of range", (unsigned long long)idx[i]); h_noff[i + 1] = h_noff[i] + (R.h_off[idx[i] + 1] - R.h_off[idx[i]]); }
    const uint64_t total = h_noff[n];
    PoolBlocks PB; void *&ds = PB.p[0], *&dq = PB.p[1], *&dof = PB.p[2]; size_t &bs = PB.b[0], &bq = PB.b[1], &bo = PB.b[2];
    HIPCHK(ctx, ngsid_pool_alloc(&ds, total + 16, &bs));
    HIPCHK(ctx, ngsid_pool_alloc(&dof, sizeof(uint64_t) * (n + 1), &bo));
    if (dev_in->qual) HIPCHK(ctx, ngsid_pool_alloc(&dq, total + 16, &bq));
    DevBuf<uint64_t> d_idx; DevBuf<unsigned long long> d_bad; HIPCHK(ctx, d_idx.alloc(n)); HIPCHK(ctx, d_bad.alloc(1));
    HIPCHK(ctx, hipMemsetAsync(d_bad.p, 0, sizeof(unsigned long long), ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dof, h_noff.data(), sizeof(uint64_t) * (n + 1), hipMemcpyHostToDevice, ctx->stream));
    if (n) HIPCHK(ctx, hipMemcpyAsync(d_idx.p, idx, sizeof(uint64_t) * n, hipMemcpyHostToDevice, ctx->stream));
    if (n) hipLaunchKernelGGL(k_reads_subset, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, ctx->stream, R.seq, R.qual, R.off, d_idx.p, (const uint64_t*)dof, n, (uint8_t*)ds, (uint8_t*)dq, d_bad.p);
    HIPCHK(ctx, hipGetLastError());
    unsigned long long hb = 0;
    HIPCHK(ctx, hipMemcpyAsync(&hb, d_bad.p, sizeof hb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (foreign) *foreign = hb;
    { std::lock_guard<std::mutex> lk(g_pool.mu); g_pool.uploads[ds] = {bs, dq, bq, dof, bo}; }
    PB.owned = false;
    dev_out->seq = (const uint8_t*)ds; dev_out->qual = (const uint8_t*)dq; dev_out->off = (const uint64_t*)dof; dev_out->n = n; dev_out->mem = NGSID_MEM_DEVICE; dev_out->_pad = 0;
    return NGSID_OK;
}

extern "C" int32_t ngsid_reads_release(ngsid_ctx* ctx, ngsid_reads_t* dev)
{
    if (!ctx) return NGSID_ERR_ARG;
    ApiClock api_clock_(ctx, "reads_release");         // the blocks go back to the cache once THIS context's streams are idle: release a read set through a context only when no call of ANOTHER context that uses it is in flight
    if (!dev || dev->mem != NGSID_MEM_DEVICE) NGSID_FAIL(ctx, NGSID_ERR_ARG, "reads_release: not a device read set");
    DevPool::Upload u;
    { std::lock_guard<std::mutex> lk(g_pool.mu); auto it = g_pool.uploads.find((void*)dev->seq); if (it == g_pool.uploads.end()) NGSID_FAIL(ctx, NGSID_ERR_ARG, "reads_release: this read set was not made by ngsid_reads_upload (or was released already)"); u = it->second; g_pool.uploads.erase(it); }
    ngsid_pool_free((void*)dev->seq, u.bs); if (u.q) ngsid_pool_free(u.q, u.bq); ngsid_pool_free(u.off, u.bo);
    dev->seq = nullptr; dev->qual = nullptr; dev->off = nullptr; dev->n = 0;
    return NGSID_OK;
}

extern "C" int32_t ngsid_ctx_option(ngsid_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return NGSID_ERR_ARG;
    static const char* known[] = {"minimizers_chunk_bases", "poa_level_budget_mb", "cluster_block", "cluster_block_cap", "cluster_trunc", "cluster_shrink_reps", "cluster_block_floor", "ed_band", "ed_win_all", "align32", "align_noclass", "poa_tiles_per_cu", "minimizers_lean", "minimizers_mode", "poa_host_levels", "align_paired", "poa_out_slots", "ed_lds_pad_kb", "ed_win6"};
    for (const char* k : known) if (!strcmp(k, name)) { ctx->options[name] = (long long)value; return NGSID_OK; }
    if (!strcmp(name, "touch")) {        // one trivial operation on the context's stream (+ wait): a caller that spends milliseconds on the host between two calls keeps the device out of its idle state
        if (ctx->mzc_fp.n < 2) HIPCHK(ctx, ctx->mzc_fp.alloc(2));
        HIPCHK(ctx, hipMemsetAsync(ctx->mzc_fp.p + 1, 0, sizeof(unsigned long long), ctx->stream)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return NGSID_OK;
    }
    if (!strcmp(name, "release_scratch")) {        // gives the context's grow-only scratch (aligner traceback, POA tiles and levels, polisher arrays) and the cached blocks back to the driver
        (void)hipStreamSynchronize(ctx->stream);
        ctx->tb.release(); ctx->bnd.release(); ctx->tb_long.release(); ctx->bnd_long.release(); ctx->aln_cls.release(); ctx->aln_psorted.release(); ctx->aln_pbin.release(); ctx->ed_tb.release(); ctx->ed_h.release(); ctx->ed_fail.release(); ctx->ed_fail2.release();
        ctx->poa_h.release(); ctx->poa_d.release(); ctx->poa_g.release(); ctx->poa_cov.release();
        for (auto& L : ctx->poa_lv) { L.out.release(); L.seqs.release(); L.job_final.release(); L.out_len.release(); L.out_span.release(); L.job_bb.release(); L.out_cw.release(); L.out_n.release(); L.out_cov.release(); L.job_off.release(); L.seq_idx.release(); L.flags.release(); L.job_list.release(); L.job_unit.release(); L.job_pos.release(); }
        ctx->mzc.valid = false; ctx->mzc_cnt.release(); ctx->mzc_hlen.release(); ctx->mz_off.release(); ctx->mz_scode.release(); ctx->mz_spos.release();
        ctx->pol_mzcode.release(); ctx->pol_mzpos.release(); ctx->pol_oseq.release(); ctx->pol_oqual.release(); ctx->pol_valid.release(); ctx->pol_bp.release(); ctx->pol_lay.release(); ctx->cl_cnt.release();
        ngsid_pool_release_all();
        return NGSID_OK;
    }
    if (!strcmp(name, "scratch_budget_mb")) {      // upper bound of the aligners' traceback scratch (default: a quarter of the free HBM at ngsid_create, at most 32 GB): several contexts on one GPU
        if (value < 64) NGSID_FAIL(ctx, NGSID_ERR_ARG, "scratch_budget_mb must be at least 64");
        ctx->scratch_budget = (size_t)value << 20; return NGSID_OK;
    }
    NGSID_FAIL(ctx, NGSID_ERR_ARG, "unknown option '%s'", name);
}

extern "C" int32_t ngsid_profile_read(ngsid_ctx* ctx, char* buf, uint64_t cap)
{
    if (!ctx || !buf || !cap) return NGSID_ERR_ARG;
    prof_collect(ctx);
    std::string out;
    for (auto& kv : ctx->prof_acc) { char line[256]; snprintf(line, sizeof line, "%s %llu %.6f\n", kv.first.c_str(), (unsigned long long)kv.second.second, kv.second.first); out += line; }
    ctx->prof_acc.clear();
    { char line[128]; snprintf(line, sizeof line, "poa_band_redo_tiles %llu 0.0\n", (unsigned long long)ctx->poa_redo_tiles); out += line; ctx->poa_redo_tiles = 0; }
    {   // not kernels: device memory handed out by the library's allocator (process wide: every context, read sets of ngsid_reads_upload included) - live now / high-water mark since the last read
        size_t live = 0, peak = 0; ngsid_pool_stats(&live, &peak, true);
        char line[160]; snprintf(line, sizeof line, "hbm_live_bytes %llu 0.0\nhbm_peak_bytes %llu 0.0\n", (unsigned long long)live, (unsigned long long)peak); out += line;
        // ... and what THIS context holds in its grow-only scratch buffers (bytes), by purpose
        size_t lv = 0; for (auto& L : ctx->poa_lv) lv += L.out.abytes + L.seqs.abytes + L.out_len.abytes + L.out_span.abytes + L.job_bb.abytes + L.out_cw.abytes + L.out_n.abytes + L.out_cov.abytes + L.job_off.abytes + L.seq_idx.abytes + L.job_list.abytes + L.job_unit.abytes + L.job_pos.abytes;
        const struct { const char* nm; size_t b; } parts[] = {
            {"mem_cluster_aligner_traceback", ctx->tb.abytes + ctx->bnd.abytes + ctx->tb_long.abytes + ctx->bnd_long.abytes + ctx->aln_cls.abytes + ctx->aln_psorted.abytes + ctx->aln_pbin.abytes + ctx->aln_pint.abytes},
            {"mem_polish_aligner_traceback", ctx->ed_tb.abytes + ctx->ed_h.abytes + ctx->ed_fail.abytes + ctx->ed_fail2.abytes},
            {"mem_poa_resident_tiles", ctx->poa_h.abytes + ctx->poa_d.abytes + ctx->poa_g.abytes + ctx->poa_cov.abytes},
            {"mem_poa_level_buffers", lv},
            {"mem_minimizers_compact", ctx->pol_mzcode.abytes + ctx->pol_mzpos.abytes + ctx->mz_off.abytes + ctx->mzc_cnt.abytes + ctx->mzc_hlen.abytes},
            {"mem_minimizers_sparse_chunk", ctx->mz_scode.abytes + ctx->mz_spos.abytes},
            {"mem_oriented_reads", ctx->pol_oseq.abytes + ctx->pol_oqual.abytes},
            {"mem_cluster_hit_matrix", ctx->cl_cnt.abytes},
            {"mem_polish_layers", ctx->pol_valid.abytes + ctx->pol_bp.abytes + ctx->pol_lay.abytes}};
        for (const auto& pt : parts) { snprintf(line, sizeof line, "%s %llu 0.0\n", pt.nm, (unsigned long long)pt.b); out += line; }
    }   // not a kernel: tiles redone with a wider band (band-edge check)
    if (ctx->stat.p) {          // work counters (not kernels): DP rows of k_poa_tile (x band columns = cell updates), DP cells of the clustering aligner
        unsigned long long h[8] = {0}; HIPCHK(ctx, hipMemcpy(h, ctx->stat.p, sizeof h, hipMemcpyDeviceToHost)); HIPCHK(ctx, hipMemset(ctx->stat.p, 0, sizeof h));
        char line[160]; snprintf(line, sizeof line, "poa_dp_rows %llu 0.0\nsg_dp_cells %llu 0.0\n", h[0], h[1]); out += line;
    }
    if (out.size() + 1 > cap) NGSID_FAIL(ctx, NGSID_ERR_CAPACITY, "profile buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return NGSID_OK;
}

// (8e step 3, boundary 8b) the merge rounds of parallel_clustering on the all-gathered representatives (parallelize.py:169-217): schedule in
// include/ngsid_merge_schedule.h, every round's clustering = ngsid_cluster_greedy on this GPU.  Identical on every rank.
static int32_t merge_cb(void* user, const ngsid_reads_t* sub, const ngsid_cluster_params_t* prm, const uint32_t* acc_rank, const int32_t* prev_batch, const double* known_err,
                        int32_t* rep_of_read, double* hpc_err_out, uint8_t* status_out)
{
    uint64_t counters[4];
    return ngsid_cluster_greedy((ngsid_ctx*)user, sub, prm, acc_rank, prev_batch, known_err, rep_of_read, hpc_err_out, status_out, counters);
}
extern "C" int32_t ngsid_merge_representatives(ngsid_ctx* ctx, const ngsid_reads_t* reps, const ngsid_cluster_params_t* prm, const uint32_t* acc_rank,
                                               const double* score, const double* hpc_err, const int32_t* batch, int32_t n_batches, int32_t* rep_of)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!reps || !prm || !rep_of || (reps->n && (!score || !batch))) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    if (reps->mem != NGSID_MEM_HOST || (reps->n && !reps->qual)) NGSID_FAIL(ctx, NGSID_ERR_ARG, "representatives are expected as a host read set with qualities (they are a few KB each)");
    return ngsid_merge_schedule(merge_cb, ctx, reps, prm, acc_rank, score, hpc_err, batch, n_batches, rep_of);
}
