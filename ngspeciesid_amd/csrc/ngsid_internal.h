// ngsid_internal.h - shared internals of libngsid_hip.so (gfx950 only; no CPU implementation exists)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <map>
#include <new>
#include <mutex>
#include <chrono>
#include <memory>
#include <time.h>
#include "../../include/ngsid.h"

// Device memory goes through a small per-process cache of freed blocks (size classes with 4 significant bits): the drivers allocate
// some forty temporaries per call, and hipFree synchronises the device and costs up to a millisecond each (a 35 ms idle gap per
// clustering call in the rocprofv3 trace).  Blocks are only handed out again after the call that freed them has synchronised its
// streams, and returning a block waits for the device like hipFree does, so reuse is safe.  The cache is released with the last context.
hipError_t ngsid_pool_alloc(void** p, size_t bytes, size_t* got);
void ngsid_pool_free(void* p, size_t bytes);
// Round 6: the context whose API call the calling thread is in (set by ApiClock for the duration of the call).  ngsid_pool_free waits for THAT context's streams only - the blocks a call
// frees were used by its own launches - instead of the whole device, so that two contexts driven by two host threads (pipeline.polish_lanes) do not serialise on each other's kernels.
struct ngsid_ctx;
extern thread_local ngsid_ctx* g_ngsid_tls_ctx;
void ngsid_pool_release_all();
int ngsid_pool_contexts();            // live contexts of this process (they share the device: budgets derived from free memory are divided by it)
size_t ngsid_pool_cached_bytes();      // bytes the cache holds for reuse (they count as used in hipMemGetInfo)
void ngsid_pool_stats(size_t* live, size_t* peak, bool reset_peak);      // bytes handed out through ngsid_pool_alloc and not returned: now / high-water mark (process wide)

// RAII device buffer (returned to the cache at scope exit; all work is synchronised before return)
template <typename T> struct DevBuf {
    T* p = nullptr; size_t n = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) ngsid_pool_free(p, abytes); }
    size_t cap = 0, abytes = 0;
    void release() { if (p) { ngsid_pool_free(p, abytes); p = nullptr; } n = 0; cap = 0; abytes = 0; }
    hipError_t alloc(size_t count) {
        if (p) { ngsid_pool_free(p, abytes); p = nullptr; }
        n = count; if (!count) count = 1;
        void* q = nullptr; hipError_t e = ngsid_pool_alloc(&q, count * sizeof(T), &abytes);
        p = (T*)q; cap = e == hipSuccess ? abytes / sizeof(T) : 0; if (e != hipSuccess) { p = nullptr; n = 0; abytes = 0; } return e;
    }
    // grow-only, contents not kept: for scratch that is reused call after call
    hipError_t reserve(size_t count) { if (p && count <= cap) { n = count; return hipSuccess; } return alloc(count + count / 8); }
    hipError_t grow(size_t count, hipStream_t s) {   // keeps contents
        if (count <= n) return hipSuccess;
        void* q = nullptr; size_t qb = 0; hipError_t e = ngsid_pool_alloc(&q, count * sizeof(T), &qb); if (e != hipSuccess) return e;
        if (p && n) { e = hipMemcpyAsync(q, p, n * sizeof(T), hipMemcpyDeviceToDevice, s); if (e != hipSuccess) return e; e = hipStreamSynchronize(s); if (e != hipSuccess) return e; }
        if (p) ngsid_pool_free(p, abytes);
        p = (T*)q; abytes = qb; cap = qb / sizeof(T); n = count; return hipSuccess;
    }
};

void* ngsid_pinned_alloc(size_t bytes);
void ngsid_pinned_free(void* p, size_t bytes);
// Host vectors in pinned memory for the big, recurring host <-> device copies.  A copy from / to pageable memory makes the runtime pin the
// pages for the transfer and release them afterwards; with 8-40 MB per hierarchy level that showed up as milliseconds on the NEXT
// submission (8 ms after the clustering results, rocprofv3 trace + host timers).  These vectors live in thread-local statics and only grow.
template <typename T> struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U> PinnedAlloc(const PinnedAlloc<U>&) {}
    // Round 6: blocks come from and go back to a process-wide cache (ngsid_pinned_alloc / _free in ngsid_api.hip: power-of-two-ish size classes, at most 4 GB kept).
    // hipHostFree waits for the whole DEVICE; with the vectors thread_local, a caller that drives a context from short-lived threads paid that wait at every growth step and
    // at every thread exit - and, with two contexts on one device, waited for the OTHER context's kernels.  The users of these vectors synchronise their stream before they touch
    // a vector again, so a block that leaves a vector is not in flight.
    T* allocate(size_t n) { void* p = ngsid_pinned_alloc((n ? n : 1) * sizeof(T)); if (!p) throw std::bad_alloc(); return (T*)p; }
    void deallocate(T* p, size_t n) { ngsid_pinned_free(p, (n ? n : 1) * sizeof(T)); }
    template <class U> bool operator==(const PinnedAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <typename T> using PinVec = std::vector<T, PinnedAlloc<T>>;

struct ProfEntry { const char* name; hipEvent_t a, b; };

typedef unsigned int ngsid_v4u_t __attribute__((ext_vector_type(4)));
struct ngsid_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char err[1024] = {0};
    size_t scratch_budget = (size_t)24 << 30;   // upper bound for traceback scratch (bytes)
    int n_cu = 256;
    DevBuf<uint64_t> tb;      // aligner traceback scratch (grow-only)
    DevBuf<int32_t> bnd;      // aligner strip boundary rows
    DevBuf<uint64_t> tb_long; DevBuf<int32_t> bnd_long;      // the same for the long-pair class of a partitioned batch (round 5: reads up to 65 535 bases)
    DevBuf<uint32_t> aln_ctr; // aligner work-queue counters (one per launch in flight) + length-class counts
    DevBuf<uint32_t> aln_cls; // pair lists of the length classes
    DevBuf<uint32_t> aln_pint, aln_psorted; DevBuf<uint8_t> aln_pbin;   // paired aligner (k_align16p.hip): bin counters / offsets, pairs sorted by bin, bin of every class entry
    DevBuf<ngsid_v4u_t> ed_tb; // traceback vectors of the edit-distance aligner
    DevBuf<int8_t> ed_h;       // its horizontal deltas between block groups
    DevBuf<uint32_t> ed_fail;  // pairs beyond the band of the first launch
    DevBuf<uint32_t> ed_fail2; // ... and beyond the wider band of the retry launch
    DevBuf<uint32_t> poa_ctr; // POA tile work-queue counter
    uint64_t poa_redo_tiles = 0;   // tiles redone with a wider band since the context was created (band-edge check)
    struct PoaLevelBufs { DevBuf<uint8_t> out, seqs /* PSeq[] */, job_final; DevBuf<int32_t> out_len, out_span, job_bb; DevBuf<uint64_t> out_cw; DevBuf<uint32_t> out_n, out_cov, job_off, seq_idx, flags, job_list, job_unit, job_pos; };
    PoaLevelBufs poa_lv[2];   // hierarchy levels ping-pong between two buffer sets (level L+1 reads what level L wrote)
    DevBuf<uint64_t> pol_mzcode; DevBuf<uint32_t> pol_mzpos; DevBuf<uint8_t> pol_oseq, pol_oqual; DevBuf<uint16_t> pol_valid; DevBuf<int32_t> pol_bp; DevBuf<uint8_t> pol_lay;   // polisher scratch (grow-only)
    DevBuf<uint64_t> cl_cnt;                                                                   // hit matrix of the clustering driver's current block (grow-only: blocks of up to 1 M items since round 6)
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};   // side streams: the launches of the small length classes overlap the big one
    hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
    void* pin = nullptr; size_t pin_bytes = 0;     // pinned host staging (device -> host copies of offsets)
    std::map<std::string, long long> options;     // ngsid_ctx_option
    bool debug_sync = false;
    // minimizers of the last clustering / polishing call (codes and positions in pol_mzcode / pol_mzpos, counts and HPC lengths here), keyed by the read set's
    // size, (k, w) and a 64-bit fingerprint of its bases and offsets: the polisher's strand detection reuses them when it is handed the reads the clustering
    // call just saw (k <= 21: one-word codes, comparable between calls) instead of running k_hpc_minimizers a second time
    struct MzCache { bool valid = false; uint64_t n = 0, total = 0; int k = 0, w = 0; unsigned long long fp = 0; } mzc;
    DevBuf<uint32_t> mzc_cnt, mzc_hlen; DevBuf<unsigned long long> mzc_fp;
    // round 5: the minimizers of a read set are kept as a COMPACT CSR (pol_mzcode / pol_mzpos indexed through mz_off[read], host mirror h_mzoff) - the kernel still
    // writes them sparsely at the reads' base offsets, but into a bounded scratch (mz_scode / mz_spos: one chunk of reads at a time, ngsid_minimizers_csr), so the
    // 12 bytes per BASE of round 1-4 (90 GB at the 10 M reads of C4) are 12 bytes per MINIMIZER (14 GB) + a fixed 3 GB
    DevBuf<uint64_t> mz_off, mz_scode; DevBuf<uint32_t> mz_spos; PinVec<uint64_t> h_mzoff;
    // host offsets of the last DEVICE-resident read set (ngsid_upload_reads): pointer, count and a 64-bit device-side fingerprint of the offsets are the key
    struct OffCache { const void* ptr = nullptr; uint64_t n = 0; unsigned long long fp = 0; std::shared_ptr<std::vector<uint64_t>> v; uint32_t maxlen = 0, minlen = 0; } offc;
    DevBuf<unsigned long long> stat;     // work counters while profiling is on (bench.py): [0] DP rows of k_poa_tile, [1] DP cells of the clustering aligner
    bool prof = false; std::vector<ProfEntry> prof_events; std::map<std::string, std::pair<double, uint64_t>> prof_acc;
    DevBuf<int32_t> poa_h; DevBuf<uint8_t> poa_d; DevBuf<uint8_t> poa_g; DevBuf<uint32_t> poa_cov;   // POA tile scratch (grow-only)
};

// Scheduling options of a context (ngsid_ctx_option): block sizes, kernel instance choice, band of the first attempt.  Set explicitly through the
// C-ABI by tests and tools - the library reads NO environment variable for them; results never depend on them (the tests run both ways).
static inline long long ngsid_opt(const ngsid_ctx* ctx, const char* name, long long dflt) { auto it = ctx->options.find(name); return it == ctx->options.end() ? dflt : it->second; }

#define NGSID_FAIL(ctx, code, ...) do { snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); return (code); } while (0)
#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); return NGSID_ERR_HIP; } } while (0)

// brackets one kernel launch with HIP events on the ctx stream when profiling is enabled
// wall time of an API call as the LIBRARY sees it (profiling on): "host_<name> <calls> <ms>" lines of ngsid_profile_read.  The caller's own clock around the same call, minus this, is
// what the binding layer adds (for a Python caller with busy worker threads: the wait for the interpreter lock when the call returns) - round 5, the CLI's sporadic stalls
struct ApiClock {
    ngsid_ctx* c; const char* nm; struct timespec t0; bool on; ngsid_ctx* prev_;
    ApiClock(ngsid_ctx* ctx, const char* name) : c(ctx), nm(name), on(ctx && ctx->prof), prev_(g_ngsid_tls_ctx) { g_ngsid_tls_ctx = ctx; if (on) clock_gettime(CLOCK_MONOTONIC, &t0); }
    ~ApiClock() { g_ngsid_tls_ctx = prev_; if (!on) return; struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1); auto& a = c->prof_acc[std::string("host_") + nm]; a.first += (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6; a.second += 1; }
};

struct ProfScope {
    ngsid_ctx* c; ProfEntry e; bool on;
    const char* dbg_name; hipStream_t st;
    ProfScope(ngsid_ctx* ctx, const char* name, hipStream_t s = nullptr) : c(ctx), on(ctx->prof), dbg_name(name), st(s ? s : ctx->stream) {
        if (c->debug_sync) { fprintf(stderr, "[ngsid] launch %s\n", name); fflush(stderr); }
        if (!on) return; e.name = name;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(e.a, st);
    }
    ~ProfScope() {
        if (c->debug_sync) { hipError_t er = hipStreamSynchronize(st); fprintf(stderr, "[ngsid] %s -> %s\n", dbg_name, hipGetErrorString(er)); fflush(stderr); }
        if (!on) return; (void)hipEventRecord(e.b, st); c->prof_events.push_back(e);
    }
};

// host copy of a read set's offsets: a shared vector, so that the copy a context keeps of the LAST device-resident read set it was handed (ngsid_ctx::offc) is
// reused by the next call on the same reads without another 8 MB download and scan (cluster -> draft consensus -> polish all see the same set)
struct OffView {
    std::shared_ptr<std::vector<uint64_t>> v;
    uint64_t& operator[](size_t i) { return (*v)[i]; }
    const uint64_t& operator[](size_t i) const { return (*v)[i]; }
    uint64_t* data() { return v->data(); }
    const uint64_t* data() const { return v->data(); }
    void resize(size_t n) { v = std::make_shared<std::vector<uint64_t>>(n); }
    size_t size() const { return v ? v->size() : 0; }
};
// A read set resident in HBM (+ host copy of the offsets, which every host-side planner needs)
struct DevReads {
    const uint8_t* seq = nullptr; const uint8_t* qual = nullptr; const uint64_t* off = nullptr;
    uint64_t n = 0, total = 0; uint32_t maxlen = 0, minlen = 0;
    OffView h_off;
    DevBuf<uint8_t> own_seq, own_qual; DevBuf<uint64_t> own_off;
};
int32_t ngsid_upload_reads(ngsid_ctx* ctx, const ngsid_reads_t* in, DevReads* out, bool need_qual);
int32_t ngsid_reads_fingerprint(ngsid_ctx* ctx, const DevReads& R, unsigned long long* fp);      // k_minimizers.hip: position-mixed 64-bit sum over the bases and the offsets (one pass, ~0.2 ms per GB)

// ---- kernels' host launchers (defined in the .hip files) ----
// per read: HPC length, minimizer count, HPC error rate, raw mean error + the minimizers (code, position in the HPC string)
// as a compact CSR (`out`; the context's own store is ngsid_ctx_mz(ctx) = pol_mzcode / pol_mzpos / mz_off / h_mzoff), computed chunk by chunk through a bounded sparse scratch.
// d_cnt / d_hlen / d_herr / d_rawerr: device arrays of R.n entries; h_cnt / h_hlen: host mirrors (R.n entries each, filled); *bad_read = index of a read with a base
// outside ACGTN or -1.  The counts are on the host when it returns; the last gather may still be in flight on the stream.
struct MzOut { DevBuf<uint64_t>* code; DevBuf<uint32_t>* pos; DevBuf<uint64_t>* off; PinVec<uint64_t>* h_off; };      // where the CSR goes (grow-only buffers; contents replaced)
static inline MzOut ngsid_ctx_mz(ngsid_ctx* ctx) { return MzOut{&ctx->pol_mzcode, &ctx->pol_mzpos, &ctx->mz_off, &ctx->h_mzoff}; }
int32_t ngsid_minimizers_csr(ngsid_ctx* ctx, const DevReads& R, int k, int w, const MzOut& out, uint32_t* d_cnt, uint32_t* d_hlen, double* d_herr, double* d_rawerr,
                             uint32_t* h_cnt, uint32_t* h_hlen, long long* bad_read);

struct AlignJob {            // device pointers
    const uint8_t* qseq; const uint64_t* qoff; const uint8_t* tseq; const uint64_t* toff;
    const uint32_t* qidx; const uint32_t* tidx; uint64_t npairs;
    int match, mismatch, ext, k; const int32_t* open; const int32_t* match_id;
    int32_t* score; int32_t* ncols; int32_t* nmatch; int32_t* region;
    // optional window break points (polish): per pair `bp_windows` records of 4 int32 {q_first,q_last,t_first,t_last}, -1 = none
    int32_t* bp; int bp_windows; int window; int32_t* span;  // span: per pair {q_begin,q_end,t_begin,t_end} of the aligned part
    // optional indirection (length classes): the k-th work item is pair pair_list[k], and the item count is read from device memory
    const uint32_t* pair_list; const uint32_t* npairs_dev;
    // optional alignment columns (int32 kernel only): ops + ops_off[p] receives one byte per column in traceback (reverse) order, 0 '=' 1 'X' 2 'I' 3 'D'
    uint8_t* ops; const uint64_t* ops_off;
    // optional SECOND index list behind the first (k_ed_align only: two query-length classes in one launch): item k >= *npairs_dev is pair_list2[k - *npairs_dev]
    const uint32_t* pair_list2; const uint32_t* npairs_dev2;
    int clip;        // k_ed_align only (round 5, aln_mode 3): != 0 = overlap-span clipping of the recorded span / break points
};
int32_t ngsid_launch_align(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, int max_open = 1 << 20, uint32_t min_qlen = 0);   // min_qlen: lower bound of the query lengths (lets empty length classes be skipped)
bool ngsid_align16_applicable(const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, int max_open);
#define NGSID_ALIGN16_MAXLEN 4000u      // sequences up to this length run the packed int16 aligners (score range)
#define NGSID_ALIGN_LONG_CLASS 5        // list / counter index of the pairs above it in a partitioned batch (aln_cls list 5, aln_ctr[13])
int32_t ngsid_launch_align16(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, uint32_t min_qlen = 0, uint32_t long_len = 0);
int32_t ngsid_side_streams(ngsid_ctx* ctx);          // creates ctx->side / events on first use
int32_t ngsid_paired_tb_words(ngsid_ctx* ctx, int cls, uint64_t npairs, uint32_t max_tlen, uint64_t* words);      // k_align16p.hip: two pairs per wave for every single-strip length class (queries of up to 896 bases)
int32_t ngsid_launch_paired_class(ngsid_ctx* ctx, const AlignJob& job, int cls, uint32_t max_tlen, hipStream_t st, uint64_t* tb);
int32_t ngsid_partition_pairs(ngsid_ctx* ctx, const AlignJob& job, uint32_t long_len = 0);      // query-length classes {<=256, <=512, <=768, <=896, rest}: lists in ctx->aln_cls, counts in ctx->aln_ctr[8..12]
int32_t ngsid_launch_ed_align(ngsid_ctx* ctx, const AlignJob& job, uint32_t max_qlen, uint32_t max_tlen, int32_t* dist_out);   // k_ed_align.hip (uses qseq..npairs, bp, bp_windows, window, span)

typedef unsigned int ngsid_v4u __attribute__((ext_vector_type(4)));
// 16-byte load served by L2 (nt): for scratch that this wave rewrites between uses, where an L1 line could be stale
__device__ __forceinline__ ngsid_v4u ngsid_load16_l2(const void* p) { return __builtin_nontemporal_load((const ngsid_v4u*)p); }

__device__ __forceinline__ int ngsid_bcode(uint8_t c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

// dev aid: NGSID_HOST_TIMERS=1 prints host-side wall time per section (stream synchronised at each mark)
struct HostTimer {
    bool on; hipStream_t st; std::chrono::steady_clock::time_point t0; const char* what;
    HostTimer(hipStream_t s, const char* w) : on(getenv("NGSID_HOST_TIMERS") != nullptr), st(s), what(w) { if (on) { (void)hipStreamSynchronize(st); t0 = std::chrono::steady_clock::now(); } }
    void mark(const char* label) { if (!on) return; (void)hipStreamSynchronize(st); auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[ngsid host] %s: %s %.2f ms   (t=%.2f)\n", what, label, std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t1.time_since_epoch()).count() - 1e3 * (double)((long long)(std::chrono::duration<double>(t1.time_since_epoch()).count()) / 1000 * 1000)); t0 = t1; }
};
