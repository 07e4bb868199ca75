// host_io.hip - (f2) host-side ingest / egress helpers of the drop-in CLI: FASTQ indexing, CSR gathers and FASTQ / FASTA / TSV record
// assembly, multi-threaded, no device work.  They replace the per-record Python of the reference's readfq / writers
// (modules/help_functions.py:13-42, modules/get_sorted_fastq_for_cluster.py:174-177, NGSpeciesID:99-120, modules/consensus.py:203-215)
// so that a million-read run spends its host time in memcpy-speed loops.  Plain C-ABI like the rest of include/ngsid.h.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <thread>
#include <system_error>
#include <string>
#include <map>
#include <deque>
#include <functional>
#include <mutex>
#include <condition_variable>
#include <sched.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/types.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <vector>
#include <algorithm>
#include <charconv>
#include <cmath>
#include "../../include/ngsid.h"

namespace {
// CPUs this process may actually use: hardware threads, the affinity mask and the container's CPU quota (cgroup v2 cpu.max / v1 cfs quota).  A GPU box shows
// 256 hardware threads to a container that may run 16: helper calls that start 32 threads each, several at a time (the CLI's background writers), exhaust the
// quota and the kernel then throttles EVERY thread of the container - the one that drives the GPU included (round 4: +0.12 s in the CLI's clustering stage).
unsigned usable_cpus()
{
    static const unsigned cached = [] {
        unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 4;
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) hw = std::min<unsigned>(hw, (unsigned)c); }
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[64] = {0}; long long per = 0; if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) { const long long c = atoll(q) / per; if (c >= 1) hw = std::min<unsigned>(hw, (unsigned)c); } fclose(f); }
        else {
            long long quota = -1, per = 0;
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
            if (quota > 0 && per > 0 && quota / per >= 1) hw = std::min<unsigned>(hw, (unsigned)(quota / per));
        }
        return std::max(1u, hw);
    }();
    return cached;
}
thread_local int t_thread_cap = 0;          // ngsid_host_thread_cap: upper bound for the helper calls of THIS thread (0 = none)
int n_threads(uint64_t work_bytes)
{
    unsigned hw = usable_cpus();
    if (const char* e = getenv("NGSID_HOST_THREADS")) { int v = atoi(e); if (v > 0) hw = (unsigned)v; }
    if (t_thread_cap > 0) hw = std::min<unsigned>(hw, (unsigned)t_thread_cap);
    const uint64_t by_size = work_bytes / (4u << 20) + 1;           // at least 4 MB per thread
    return (int)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw, 32u), by_size));
}
// Process-wide bound on the helper threads that RUN at the same time (round 5): every call splits its work into T ranges as before (the partition, and with it every per-range
// array of the callers, is unchanged), but a range of a background caller only starts once it holds one of usable_cpus() - 3 slots.  The CLI's eight background writers with up to eight helper threads each
// could put 40+ runnable threads into a 16-CPU quota; an exhausted CFS period stops every thread of the container, the one that drives the GPU included (bench.py
// config.cli.cgroup_cpu_during_the_leg: one throttled period per leg without the bound, none with it).
struct HelperSlots {
    std::mutex m; std::condition_variable cv; int free_;
    HelperSlots() { int v = (int)usable_cpus() - 3; if (const char* e = getenv("NGSID_HOST_SLOTS")) { const int x = atoi(e); if (x > 0) v = x; } free_ = std::max(2, v); }
    void acquire() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return free_ > 0; }); --free_; }
    void release() { { std::lock_guard<std::mutex> l(m); ++free_; } cv.notify_one(); }
};
HelperSlots& helper_slots() { static HelperSlots s; return s; }
template <class F> void parallel_ranges(uint64_t n, int T, F f)
{
    if (T <= 1 || n < 2) { f(0, n, 0); return; }
    // the bound applies to BACKGROUND callers (threads that took a cap with ngsid_host_thread_cap: the CLI's writers and gather workers); the thread that drives the GPU never has
    // helper calls of its own in flight while it launches, so its calls run unbounded and three CPUs of the quota stay free for it and the runtime's threads
    HelperSlots& S = helper_slots(); const bool bg = t_thread_cap > 0;
    std::vector<std::thread> th; th.reserve(T);
    for (int t = 0; t < T; ++t) { const uint64_t a = n * t / T, b = n * (t + 1) / T; th.emplace_back([=, &S] { if (bg) S.acquire(); f(a, b, t); if (bg) S.release(); }); }      // (the ranges are leaves: no helper call inside f)
    for (auto& x : th) x.join();
}
}  // namespace

// upper bound of the worker threads the ngsid_host_* helpers start when called from the CALLING thread (0 = no bound): background writers take a few, so that
// the thread that drives the GPU keeps a core.  Returns the previous bound.
extern "C" int32_t ngsid_host_thread_cap(int32_t n) { const int old = t_thread_cap; if (n >= 0) t_thread_cap = n; return old; }

// Index of a plain 4-line FASTQ held in memory.  Pass 1 (rec == NULL): counts the lines, checks the structure, returns the number of
// records in *n_records.  Pass 2: fills, per record r, rec[4r..4r+3] = offsets of the name (after '@'), the sequence, the '+' line and the
// quality string, and name_len / seq_len.  Returns 0 = ok, 1 = not a plain 4-line FASTQ (multi-line records, FASTA, truncated file, a
// quality string whose length differs ...): the caller then uses the general reader.
extern "C" int32_t ngsid_host_fastq_index(const uint8_t* buf, uint64_t len, uint64_t* rec, uint32_t* name_len, uint32_t* seq_len,
                                          uint64_t cap_records, uint64_t* n_records)
{
    if (!buf || !n_records) return NGSID_ERR_ARG;
    const int T = n_threads(len);
    // the caller counts first (rec == NULL) and indexes second: the per-range line counts of the counting call are kept for the indexing call on the same buffer (same address,
    // length, thread count and the same first / last 64 bytes), which saves one of three passes over the file image (1.5 GB at C3)
    struct CountCache { const uint8_t* buf = nullptr; uint64_t len = 0; int T = 0; uint64_t sig = 0; std::vector<uint64_t> cnt; };
    static thread_local CountCache cc;
    auto signature = [&]() { uint64_t h = len * 0x9E3779B97F4A7C15ull; const uint64_t m = len < 64 ? len : 64; for (uint64_t i = 0; i < m; ++i) { h = (h ^ buf[i]) * 0x100000001B3ull; h = (h ^ buf[len - 1 - i]) * 0x100000001B3ull; } return h; };
    const uint64_t sig = signature();
    std::vector<uint64_t> cnt(T + 1, 0);
    if (rec && cc.buf == buf && cc.len == len && cc.T == T && cc.sig == sig && cc.cnt.size() == (size_t)T + 1) { cnt = cc.cnt; cc.buf = nullptr; cc.cnt.clear(); }      // single use: a caller that refills the same buffer and indexes again without counting gets a fresh count
    else {
        parallel_ranges(len, T, [&](uint64_t a, uint64_t b, int t) { uint64_t c = 0; const uint8_t* p = buf + a; const uint8_t* e = buf + b;
            while (p < e) { const uint8_t* q = (const uint8_t*)memchr(p, '\n', (size_t)(e - p)); if (!q) break; ++c; p = q + 1; } cnt[t + 1] = c; });
        for (int t = 0; t < T; ++t) cnt[t + 1] += cnt[t];
        if (!rec) { cc.buf = buf; cc.len = len; cc.T = T; cc.sig = sig; cc.cnt = cnt; }
    }
    uint64_t nlines = cnt[T];
    const bool tail = len > 0 && buf[len - 1] != '\n';            // last line without a newline
    if (tail) ++nlines;
    if (nlines % 4 != 0) return 1;
    *n_records = nlines / 4;
    if (!rec) return 0;
    if (cap_records < nlines / 4) return NGSID_ERR_CAPACITY;
    // line starts: thread t writes the starts of the lines that BEGIN after a newline found in its range
    std::vector<uint64_t> starts(nlines + 1);
    starts[0] = 0;
    parallel_ranges(len, T, [&](uint64_t a, uint64_t b, int t) { uint64_t k = cnt[t] + 1; const uint8_t* p = buf + a; const uint8_t* e = buf + b;
        while (p < e) { const uint8_t* q = (const uint8_t*)memchr(p, '\n', (size_t)(e - p)); if (!q) break; if (k <= nlines) starts[k] = (uint64_t)(q + 1 - buf); ++k; p = q + 1; } });
    if (tail) starts[nlines] = len + 1;                           // virtual newline behind the last line
    const uint64_t nr = nlines / 4;
    // one shared flag (the record pass may run with MORE threads than the line passes - short records - so a per-thread array sized by T was
    // overrun and lost the error); a bad record does not stop its thread: every other record of the range is still indexed
    std::atomic<int> bad{0};
    // length of the line [s, nx - 1) without its line terminator: "\n" or "\r\n" (the reference opens the file in text mode, which drops the '\r')
    auto linelen = [&](uint64_t s, uint64_t nx) -> uint64_t { uint64_t e = nx - 1; if (e > s && e - 1 < len && buf[e - 1] == '\r') --e; return e - s; };
    parallel_ranges(nr, n_threads(nr * 64), [&](uint64_t a, uint64_t b, int) {
        for (uint64_t r = a; r < b; ++r) {
            const uint64_t s0 = starts[4 * r], s1 = starts[4 * r + 1], s2 = starts[4 * r + 2], s3 = starts[4 * r + 3], s4 = starts[4 * r + 4];
            if (s0 >= len || buf[s0] != '@' || s2 >= len || buf[s2] != '+') { bad.store(1, std::memory_order_relaxed); continue; }
            const uint64_t sl = linelen(s1, s2), ql = linelen(s3, s4), nl = linelen(s0 + 1, s1);
            if (sl != ql || sl > 0xffffffffull || nl > 0xffffffffull) { bad.store(1, std::memory_order_relaxed); continue; }
            rec[4 * r] = s0 + 1; rec[4 * r + 1] = s1; rec[4 * r + 2] = s2; rec[4 * r + 3] = s3;
            name_len[r] = (uint32_t)nl; seq_len[r] = (uint32_t)sl;
        } });
    return bad.load() ? 1 : 0;
}

// dst[dst_off[i] .. +len[i]) = src[src_off[i] .. +len[i]) for i < n  (CSR gather / scatter of variable-length byte records)
extern "C" int32_t ngsid_host_gather(const uint8_t* src, const uint64_t* src_off, const uint32_t* len, uint64_t n, uint8_t* dst, const uint64_t* dst_off)
{
    if ((!src || !dst || !src_off || !dst_off || !len) && n) return NGSID_ERR_ARG;
    uint64_t total = 0; for (uint64_t i = 0; i < n; ++i) total += len[i];
    parallel_ranges(n, n_threads(total), [&](uint64_t a, uint64_t b, int) { for (uint64_t i = a; i < b; ++i) memcpy(dst + dst_off[i], src + src_off[i], len[i]); });
    return NGSID_OK;
}

// In place: upper-case a..z, everything that is then not one of A C G T N becomes N.  Returns the number of bytes changed in *changed
// (soft-masked / IUPAC / U bases: the reference clusters them as literal characters, this build as N - see DESIGN.md).
extern "C" int32_t ngsid_host_normalize_bases(uint8_t* seq, uint64_t len, uint64_t* changed)
{
    if (!seq && len) return NGSID_ERR_ARG;
    const int T = n_threads(len); std::vector<uint64_t> c(T, 0);
    parallel_ranges(len, T, [&](uint64_t a, uint64_t b, int t) { uint64_t k = 0;
        for (uint64_t i = a; i < b; ++i) { uint8_t ch = seq[i]; if (ch >= 'a' && ch <= 'z') ch = (uint8_t)(ch - 32);
            if (ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N') ch = 'N'; if (ch != seq[i]) { seq[i] = ch; ++k; } } c[t] = k; });
    uint64_t tot = 0; for (auto v : c) tot += v; if (changed) *changed = tot;
    return NGSID_OK;
}

// pos[i] = number of j < i with rep[j] == rep[i]: the position of read i in its cluster's read list when the list is the representative followed by the
// joining reads in processing order (cluster.py:338-345 on one pass over score-sorted reads).  rep values are read indices < n.  One serial pass.
extern "C" int32_t ngsid_host_list_positions(const int64_t* rep, uint64_t n, int64_t* pos)
{
    if ((!rep || !pos) && n) return NGSID_ERR_ARG;
    std::vector<uint32_t> cnt(n, 0);
    for (uint64_t i = 0; i < n; ++i) { const int64_t r = rep[i]; if (r < 0 || (uint64_t)r >= n) return NGSID_ERR_ARG; pos[i] = (int64_t)cnt[(size_t)r]++; }
    return NGSID_OK;
}

// Clusters from the representative map: reps[0..*n_reps) = the reads with rep[i] == i in ascending order (= dense cluster ids), counts[c] = size of cluster c,
// grp_off[0..*n_reps] = their prefix sums, order[grp_off[c] .. grp_off[c+1]) = the reads of cluster c in ascending read index (a stable sort of the reads by
// cluster: the representative first when it is the cluster's smallest index, the members in processing order - cluster.py:338-345).  reps / counts need room
// for n entries, grp_off for n + 1.  Three counting passes instead of NumPy's compare + cumsum + gather + radix argsort + bincount.
template <typename RT>
static int32_t group_by_rep_impl(const RT* rep, uint64_t n, int64_t* reps, uint64_t* n_reps, uint32_t* order, uint64_t* grp_off, int64_t* counts)
{
    if ((!rep || !reps || !order || !grp_off || !counts) && n) return NGSID_ERR_ARG;
    if (!n_reps || n > 0xffffffffull) return NGSID_ERR_ARG;
    std::vector<uint32_t> dense(n);
    uint64_t R = 0;
    for (uint64_t i = 0; i < n; ++i) { const int64_t r = (int64_t)rep[i]; if (r < 0 || (uint64_t)r >= n) return NGSID_ERR_ARG; dense[i] = (uint32_t)R; if ((uint64_t)r == i) { reps[R] = (int64_t)i; counts[R] = 0; ++R; } }
    for (uint64_t i = 0; i < n; ++i) { const uint64_t r = (uint64_t)rep[i]; if ((uint64_t)rep[r] != r) return NGSID_ERR_ARG; ++counts[dense[r]]; }      // (a representative represents itself)
    std::vector<uint64_t> cur(R + 1);
    grp_off[0] = 0; for (uint64_t c = 0; c < R; ++c) { grp_off[c + 1] = grp_off[c] + (uint64_t)counts[c]; cur[c] = grp_off[c]; }
    for (uint64_t i = 0; i < n; ++i) order[cur[dense[(uint64_t)rep[i]]]++] = (uint32_t)i;
    *n_reps = R;
    return NGSID_OK;
}
extern "C" int32_t ngsid_host_group_by_rep(const int64_t* rep, uint64_t n, int64_t* reps, uint64_t* n_reps, uint32_t* order, uint64_t* grp_off, int64_t* counts)
{ return group_by_rep_impl(rep, n, reps, n_reps, order, grp_off, counts); }
// the same on the int32 map ngsid_cluster_greedy returns (no widening copy between the clustering call and the consensus call: the GPU idles while the host groups)
extern "C" int32_t ngsid_host_group_by_rep32(const int32_t* rep, uint64_t n, int64_t* reps, uint64_t* n_reps, uint32_t* order, uint64_t* grp_off, int64_t* counts)
{ return group_by_rep_impl(rep, n, reps, n_reps, order, grp_off, counts); }

// order = the stable argsort of v in DESCENDING order (list.sort(key=score, reverse=True) of get_sorted_fastq_for_cluster.py:174: equal scores keep their
// input order; -0.0 == 0.0; NaNs last, in input order).  LSD radix sort on the order-reversing bit pattern, 11-bit digits, digits that are equal for all keys
// skipped, per-thread histograms (stable: thread t owns a contiguous range of the current order).
extern "C" int32_t ngsid_host_argsort_desc(const double* v, uint64_t n, uint64_t* order)
{
    if ((!v || !order) && n) return NGSID_ERR_ARG;
    if (n == 0) return NGSID_OK;
    std::vector<uint64_t> key(n), key2(n), ord2(n);
    const int T = n_threads(n * 16);
    std::vector<uint64_t> orv(T, 0), andv(T, ~0ull);
    parallel_ranges(n, T, [&](uint64_t a, uint64_t b, int t) { uint64_t o = 0, an = ~0ull;
        for (uint64_t i = a; i < b; ++i) {
            double x = v[i]; uint64_t u;
            if (x != x) u = ~0ull;                                        // NaN: after everything else
            else { x = -x + 0.0; memcpy(&u, &x, 8); u = (u >> 63) ? ~u : (u | 0x8000000000000000ull); if (u == ~0ull) u = ~0ull - 1; }   // ascending order of -x
            key[i] = u; order[i] = i; o |= u; an &= u; }
        orv[t] = o; andv[t] = an; });
    uint64_t o = 0, an = ~0ull; for (int t = 0; t < T; ++t) { o |= orv[t]; an &= andv[t]; }
    const uint64_t differ = o ^ an;                                       // bits in which at least two keys differ
    uint64_t* k0 = key.data(); uint64_t* k1 = key2.data(); uint64_t* o0 = order; uint64_t* o1 = ord2.data();
    constexpr int B = 11, R = 1 << B;
    std::vector<uint64_t> hist((size_t)T * R);
    for (int sh = 0; sh < 64; sh += B) {
        if (((differ >> sh) & (R - 1)) == 0) continue;
        std::fill(hist.begin(), hist.end(), 0);
        parallel_ranges(n, T, [&](uint64_t a, uint64_t b, int t) { uint64_t* h = hist.data() + (size_t)t * R; for (uint64_t i = a; i < b; ++i) ++h[(k0[i] >> sh) & (R - 1)]; });
        uint64_t run = 0;
        for (int d = 0; d < R; ++d) for (int t = 0; t < T; ++t) { uint64_t& h = hist[(size_t)t * R + d]; const uint64_t c = h; h = run; run += c; }
        parallel_ranges(n, T, [&](uint64_t a, uint64_t b, int t) { uint64_t* h = hist.data() + (size_t)t * R;
            for (uint64_t i = a; i < b; ++i) { const uint64_t dst = h[(k0[i] >> sh) & (R - 1)]++; k1[dst] = k0[i]; o1[dst] = o0[i]; } });
        std::swap(k0, k1); std::swap(o0, o1);
    }
    if (o0 != order) memcpy(order, o0, 8 * n);
    return NGSID_OK;
}

// Record writer.  Output record j (j < n) is built from read idx[j]:
//   kind 0 (FASTQ):  '@' name sfx_j '\n' seq '\n' '+' '\n' qual '\n'
//   kind 1 (TSV):    pre_j '\t' name '\n'                                  (final_clusters.tsv: pre_j = the cluster's output id)
// name = names[name_off[i] .. +name_len[i]) truncated at the first white space when first_token != 0; sfx / pre are CSR strings indexed by j
// (sfx_off == NULL: none), or by the read index idx[j] when sfx_by_read != 0.  The file is created (append == 0) or appended to.  Returns NGSID_ERR_ARG when the file cannot be written.
static int32_t write_records_impl(const char* path, int32_t append, int32_t kind, uint64_t n, const uint64_t* idx,
                                  const uint8_t* names, const uint64_t* name_off, const uint32_t* name_len, int32_t first_token,
                                  const uint8_t* sfx, const uint64_t* sfx_off, int32_t sfx_by_read,
                                  const uint8_t* seq, const uint8_t* qual, const uint64_t* off)
{
    if (!path || (n && (!idx || !names || !name_off || !name_len))) return NGSID_ERR_ARG;
    if (kind == 0 && n && (!seq || !qual || !off)) return NGSID_ERR_ARG;
    // Two passes: record sizes -> file offsets (prefix sum), then every worker thread assembles its records piece by piece and writes each piece with pwrite() at
    // its own offset (round 4: one fwrite() per 65 536-record chunk was a serial 1.5 GB copy into the page cache, 0.45 s of the CLI's 1.4 s at C3)
    if (getenv("NGSID_WRITE_TRACE")) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[write] %s: start %.3f\n", path, ts.tv_sec % 1000 + ts.tv_nsec / 1e9); }
    const int fd = open(path, O_WRONLY | O_CREAT | (append ? 0 : O_TRUNC), 0666); if (fd < 0) return NGSID_ERR_ARG;       // (0666 & ~umask, like fopen; append mode takes its base offset once: the caller must be the only writer of the file)
    const off_t base = append ? lseek(fd, 0, SEEK_END) : 0;
    if (base < 0) { close(fd); return NGSID_ERR_ARG; }
    std::vector<uint64_t> roff(n + 1, 0);
    auto name_length = [&](uint64_t i) -> uint32_t {
        uint32_t L = name_len[i];
        if (first_token) { const uint8_t* p = names + name_off[i]; uint32_t k = 0; while (k < L && p[k] != ' ' && !(p[k] >= 9 && p[k] <= 13)) ++k;      /* str.split() white space */ L = k; }
        return L; };
    parallel_ranges(n, n_threads(n * 64), [&](uint64_t a, uint64_t b, int) { for (uint64_t j = a; j < b; ++j) {
        const uint64_t i = idx[j]; const uint32_t L = name_length(i);
        const uint64_t sj = sfx_by_read ? i : j;
        const uint64_t sl = sfx_off ? sfx_off[sj + 1] - sfx_off[sj] : 0;
        roff[j + 1] = kind == 0 ? 1 + L + sl + 1 + 2 * (off[i + 1] - off[i]) + 1 + 2 + 1 : sl + 1 + L + 1; } });
    for (uint64_t j = 0; j < n; ++j) roff[j + 1] += roff[j];
    const uint64_t total = roff[n];
    std::atomic<int> failed{0};
    if (total) {
        const int T = n_threads(total);
        parallel_ranges(n, T, [&](uint64_t a, uint64_t b, int) {
            std::vector<uint8_t> out; const uint64_t PIECE = 8u << 20;
            uint64_t j0 = a;
            while (j0 < b && !failed.load(std::memory_order_relaxed)) {
                uint64_t j1 = j0; while (j1 < b && roff[j1 + 1] - roff[j0] <= PIECE) ++j1;
                if (j1 == j0) j1 = j0 + 1;                              // a single record larger than a piece
                out.resize(roff[j1] - roff[j0]);
                for (uint64_t j = j0; j < j1; ++j) {
                    const uint64_t i = idx[j]; uint8_t* o = out.data() + (roff[j] - roff[j0]); const uint32_t L = name_length(i);
                    const uint64_t sj = sfx_by_read ? i : j;
                    const uint64_t sl = sfx_off ? sfx_off[sj + 1] - sfx_off[sj] : 0;
                    if (kind == 0) {
                        const uint64_t l = off[i + 1] - off[i];
                        *o++ = '@'; memcpy(o, names + name_off[i], L); o += L; if (sl) { memcpy(o, sfx + sfx_off[sj], sl); o += sl; } *o++ = '\n';
                        memcpy(o, seq + off[i], l); o += l; *o++ = '\n'; *o++ = '+'; *o++ = '\n'; memcpy(o, qual + off[i], l); o += l; *o++ = '\n';
                    } else {
                        if (sl) { memcpy(o, sfx + sfx_off[sj], sl); o += sl; } *o++ = '\t'; memcpy(o, names + name_off[i], L); o += L; *o++ = '\n';
                    }
                }
                uint64_t done = 0; const uint64_t len = out.size();
                while (done < len) { const ssize_t w = pwrite(fd, out.data() + done, (size_t)(len - done), base + (off_t)(roff[j0] + done)); if (w <= 0) { failed.store(1); break; } done += (uint64_t)w; }
                j0 = j1;
            } });
    }
    if (getenv("NGSID_WRITE_TRACE")) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); fprintf(stderr, "[write] %s: %llu records, %.1f MB, cap %d, end %.3f\n", path, (unsigned long long)n, total / 1e6, t_thread_cap, ts.tv_sec % 1000 + ts.tv_nsec / 1e9); }
    int32_t rc = failed.load() ? NGSID_ERR_ARG : NGSID_OK;
    if (rc != NGSID_OK) (void)!ftruncate(fd, base);           // a failed write leaves the file as it was found, not a sparse image at its full length
    if (close(fd) != 0) rc = NGSID_ERR_ARG;
    return rc;
}
extern "C" int32_t ngsid_host_write_records(const char* path, int32_t append, int32_t kind, uint64_t n, const uint64_t* idx,
                                            const uint8_t* names, const uint64_t* name_off, const uint32_t* name_len, int32_t first_token,
                                            const uint8_t* sfx, const uint64_t* sfx_off, int32_t sfx_by_read,
                                            const uint8_t* seq, const uint8_t* qual, const uint64_t* off)
{
    return write_records_impl(path, append, kind, n, idx, names, name_off, name_len, first_token, sfx, sfx_off, sfx_by_read, seq, qual, off);
}

// ---- native background writers (round 5).  A caller whose host language has an interpreter lock (the Python CLI) used to run its writers as interpreter threads: every return of the
// launch thread from a library call then competed with up to eight of them for the lock (30 - 110 ms waits in about a quarter of the CLI runs, profiles/NOTES.md).  The queue below
// takes a record-writer job and returns at once; eight native worker threads run the jobs (as background callers: their helper threads count against the process-wide bound above).
// Every pointer of a job must stay valid until the job was waited for.
namespace {
struct AsyncJobs {
    struct Job { std::function<int32_t()> fn; int32_t rc = 0; bool done = false; int threads = 8; };
    std::mutex m; std::condition_variable cv_work, cv_done; std::deque<uint64_t> q; std::map<uint64_t, Job> jobs; uint64_t next_id = 1; std::vector<std::thread> workers; bool stop = false;
    void start() { if (!workers.empty()) return; for (int i = 0; i < 8; ++i) workers.emplace_back([this] { run(); }); }
    void run() {
        for (;;) {
            uint64_t id; std::function<int32_t()> fn;
            { std::unique_lock<std::mutex> l(m); cv_work.wait(l, [&] { return stop || !q.empty(); }); if (stop && q.empty()) return; id = q.front(); q.pop_front(); fn = jobs[id].fn; t_thread_cap = jobs[id].threads; }      // (> 0: a background caller, bounded process-wide)
            const int32_t rc = fn();
            { std::lock_guard<std::mutex> l(m); Job& j = jobs[id]; j.rc = rc; j.done = true; j.fn = nullptr; }
            cv_done.notify_all();
        }
    }
    uint64_t submit(std::function<int32_t()> fn, int threads) { std::lock_guard<std::mutex> l(m); start(); const uint64_t id = next_id++; jobs[id].fn = std::move(fn); jobs[id].threads = threads > 0 ? threads : 8; q.push_back(id); cv_work.notify_one(); return id; }
    int32_t wait(uint64_t id) { std::unique_lock<std::mutex> l(m); auto it = jobs.find(id); if (it == jobs.end()) return NGSID_ERR_ARG; cv_done.wait(l, [&] { return jobs[id].done; }); const int32_t rc = jobs[id].rc; jobs.erase(id); return rc; }
    ~AsyncJobs() { { std::lock_guard<std::mutex> l(m); stop = true; } cv_work.notify_all(); for (auto& w : workers) if (w.joinable()) w.join(); }
};
AsyncJobs& async_jobs() { static AsyncJobs* a = new AsyncJobs(); return *a; }       // (never destroyed: worker threads must not be joined from a static destructor at interpreter exit)
}  // namespace
extern "C" int32_t ngsid_host_write_records_async(const char* path, int32_t append, int32_t kind, uint64_t n, const uint64_t* idx,
                                                  const uint8_t* names, const uint64_t* name_off, const uint32_t* name_len, int32_t first_token,
                                                  const uint8_t* sfx, const uint64_t* sfx_off, int32_t sfx_by_read,
                                                  const uint8_t* seq, const uint8_t* qual, const uint64_t* off, int32_t max_threads, uint64_t* job)
{
    if (!path || !job) return NGSID_ERR_ARG;
    try {
        std::string p(path);
        *job = async_jobs().submit([=] { return write_records_impl(p.c_str(), append, kind, n, idx, names, name_off, name_len, first_token, sfx, sfx_off, sfx_by_read, seq, qual, off); }, max_threads);
        return NGSID_OK;
    } catch (...) { return NGSID_ERR_ARG; }          // (no worker thread / no memory: the caller falls back to the synchronous writer or reports it)
}
extern "C" int32_t ngsid_host_async_wait(uint64_t job) { return async_jobs().wait(job); }

// (a17, boundary 8b(1)) read_alignments_it_{i}.paf of run_racon (consensus.py:112-121: `minimap2 -x map-ont center reads` -> 12-column PAF without CIGAR).  Line j (j < n) is read
// idx[j] with the alignment record aln[6 j ..] of ngsid_polish_trace_aln = {strand, q_begin, q_end, t_begin, t_end, distance}; a record with strand < 0 writes no line (minimap2
// lists mapped reads only).  Columns: query name = first white-space token of name + suffix (the header the pooled read file carries, consensus.py:213), query length, q_begin,
// q_end, '+' / '-', target name, target length, t_begin, t_end, residue matches, alignment block length, mapping quality.  The polisher aligns by edit distance and keeps no
// chain: block length = the longer of the two spans, matches = block length - distance (not below 0), mapping quality 255 (= not available, as the PAF format defines it).
static int32_t write_paf_impl(const char* path, uint64_t n, const uint64_t* idx, const uint8_t* names, const uint64_t* name_off, const uint32_t* name_len,
                              const uint8_t* sfx, const uint64_t* sfx_off, const uint64_t* off, const int32_t* aln, const char* tname, uint32_t tlen)
{
    if (!path || !tname || (n && (!idx || !names || !name_off || !name_len || !off || !aln))) return NGSID_ERR_ARG;
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666); if (fd < 0) return NGSID_ERR_ARG;
    const size_t tl = strlen(tname);
    const int T = std::max(1, n_threads(n * 96));
    std::vector<std::vector<uint8_t>> part((size_t)T);
    // raw-pointer formatting into a buffer sized for the worst case of the range (a line = name + suffix + target name + 9 numbers of at most 20 digits + 12 separators)
    auto put_u = [](uint8_t* o, uint64_t v) -> uint8_t* { char b[24]; int k = 0; do { b[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k) *o++ = (uint8_t)b[--k]; return o; };
    parallel_ranges(n, T, [&](uint64_t a, uint64_t b, int t) {
        size_t cap = 0;
        for (uint64_t j = a; j < b; ++j) { if (aln[6 * j] < 0) continue; const uint64_t i = idx[j]; cap += (size_t)name_len[i] + (sfx_off ? (size_t)(sfx_off[i + 1] - sfx_off[i]) : 0) + tl + 9 * 20 + 16; }
        std::vector<uint8_t>& ov = part[(size_t)t]; ov.resize(cap); uint8_t* o = ov.data();
        for (uint64_t j = a; j < b; ++j) {
            const int32_t* r = aln + 6 * j; if (r[0] < 0) continue;
            const uint64_t i = idx[j]; const uint8_t* nm = names + name_off[i]; uint32_t L = name_len[i], k = 0;
            while (k < L && nm[k] != ' ' && !(nm[k] >= 9 && nm[k] <= 13)) ++k;
            memcpy(o, nm, k); o += k;
            if (k == L && sfx_off) { const size_t sl = (size_t)(sfx_off[i + 1] - sfx_off[i]); memcpy(o, sfx + sfx_off[i], sl); o += sl; }        // (a name cut at a blank loses the suffix behind it: str.split()[0] of name + suffix)
            const uint64_t qs = (uint64_t)r[1], qe = (uint64_t)r[2], ts = (uint64_t)r[3], te = (uint64_t)r[4];
            const uint64_t blk = std::max(qe - qs, te - ts), nm_ = r[5] >= 0 ? (blk > (uint64_t)r[5] ? blk - (uint64_t)r[5] : 0) : std::min(qe - qs, te - ts);
            *o++ = '\t'; o = put_u(o, off[i + 1] - off[i]); *o++ = '\t'; o = put_u(o, qs); *o++ = '\t'; o = put_u(o, qe); *o++ = '\t'; *o++ = r[0] ? '-' : '+'; *o++ = '\t';
            memcpy(o, tname, tl); o += tl; *o++ = '\t'; o = put_u(o, tlen); *o++ = '\t'; o = put_u(o, ts); *o++ = '\t'; o = put_u(o, te); *o++ = '\t';
            o = put_u(o, nm_); *o++ = '\t'; o = put_u(o, blk); *o++ = '\t'; *o++ = '2'; *o++ = '5'; *o++ = '5'; *o++ = '\n';
        }
        ov.resize((size_t)(o - ov.data())); });
    std::vector<uint64_t> base((size_t)T + 1, 0); for (int t = 0; t < T; ++t) base[t + 1] = base[t] + part[(size_t)t].size();
    std::atomic<int> failed{0};
    parallel_ranges((uint64_t)T, T, [&](uint64_t a, uint64_t b, int) { for (uint64_t t = a; t < b; ++t) {
        const std::vector<uint8_t>& o = part[(size_t)t]; uint64_t done = 0;
        while (done < o.size()) { const ssize_t w = pwrite(fd, o.data() + done, (size_t)(o.size() - done), (off_t)(base[t] + done)); if (w <= 0) { failed.store(1); break; } done += (uint64_t)w; } } });
    int32_t rc = failed.load() ? NGSID_ERR_ARG : NGSID_OK;
    if (close(fd) != 0) rc = NGSID_ERR_ARG;
    return rc;
}
// job == NULL: synchronous; else queued on the native background writers (every pointer must stay valid until ngsid_host_async_wait(*job) returned; the path and the target name are copied)
extern "C" int32_t ngsid_host_write_paf(const char* path, uint64_t n, const uint64_t* idx, const uint8_t* names, const uint64_t* name_off, const uint32_t* name_len,
                                        const uint8_t* sfx, const uint64_t* sfx_off, const uint64_t* off, const int32_t* aln, const char* tname, uint32_t tlen, int32_t max_threads, uint64_t* job)
{
    if (!job) return write_paf_impl(path, n, idx, names, name_off, name_len, sfx, sfx_off, off, aln, tname, tlen);
    if (!path || !tname) return NGSID_ERR_ARG;
    try {
        std::string p(path), tn(tname);
        *job = async_jobs().submit([=] { return write_paf_impl(p.c_str(), n, idx, names, name_off, name_len, sfx, sfx_off, off, aln, tn.c_str(), tlen); }, max_threads);
        return NGSID_OK;
    } catch (...) { return NGSID_ERR_ARG; }
}

// Read-only mapping of an input file and its release (round 5).  Unmapping the 1.5 GB FASTQ of C3 takes ~75 ms (page-table teardown); done by the interpreter's own mmap object it
// happened on the launch thread, with the interpreter lock held, in the middle of the ingest.  ngsid_host_unmap_file(..., 1) hands it to a detached native thread.
extern "C" int32_t ngsid_host_map_file(const char* path, const uint8_t** data, uint64_t* len)
{
    if (!path || !data || !len) return NGSID_ERR_ARG;
    *data = nullptr; *len = 0;
    const int fd = open(path, O_RDONLY); if (fd < 0) return NGSID_ERR_ARG;
    struct stat st; if (fstat(fd, &st) != 0) { close(fd); return NGSID_ERR_ARG; }
    if (st.st_size == 0) { close(fd); return NGSID_OK; }
    void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return NGSID_ERR_ARG;
    *data = (const uint8_t*)p; *len = (uint64_t)st.st_size;
    return NGSID_OK;
}
extern "C" int32_t ngsid_host_unmap_file(const uint8_t* data, uint64_t len, int32_t in_background)
{
    if (!data || !len) return NGSID_OK;
    if (in_background) {
        try { std::thread([=] { (void)munmap((void*)data, (size_t)len); }).detach(); return NGSID_OK; }
        catch (...) { /* no thread to be had: unmap here */ }
    }
    return munmap((void*)data, (size_t)len) == 0 ? NGSID_OK : NGSID_ERR_ARG;
}

// decimal strings of n integers as a CSR (buf: sum of the digit counts, off: n + 1) - the output ids of final_clusters.tsv; *needed = bytes of buf (call with cap = 0 to size)
extern "C" int32_t ngsid_host_int_prefixes(const int64_t* v, uint64_t n, uint8_t* buf, uint64_t cap, uint64_t* off, uint64_t* needed)
{
    if ((n && !v) || !off || !needed) return NGSID_ERR_ARG;
    auto ndig = [](int64_t x) { uint64_t u = x < 0 ? (uint64_t)(-(x + 1)) + 1 : (uint64_t)x; int d = 1; while (u >= 10) { u /= 10; ++d; } return d + (x < 0 ? 1 : 0); };
    off[0] = 0;
    parallel_ranges(n, n_threads(n * 16), [&](uint64_t a, uint64_t b, int) { for (uint64_t i = a; i < b; ++i) off[i + 1] = (uint64_t)ndig(v[i]); });
    for (uint64_t i = 0; i < n; ++i) off[i + 1] += off[i];
    *needed = off[n];
    if (cap < off[n] || (off[n] && !buf)) return cap == 0 ? NGSID_OK : NGSID_ERR_ARG;
    parallel_ranges(n, n_threads(n * 16), [&](uint64_t a, uint64_t b, int) { for (uint64_t i = a; i < b; ++i) {
        uint8_t* e = buf + off[i + 1]; int64_t x = v[i]; uint64_t u = x < 0 ? (uint64_t)(-(x + 1)) + 1 : (uint64_t)x;
        do { *--e = (uint8_t)('0' + u % 10); u /= 10; } while (u);
        if (x < 0) *--e = '-'; } });
    return NGSID_OK;
}

// Number of bases that ngsid_host_normalize_bases would change (no copy needed when it is 0).
extern "C" int32_t ngsid_host_count_foreign_bases(const uint8_t* seq, uint64_t len, uint64_t* count)
{
    if ((!seq && len) || !count) return NGSID_ERR_ARG;
    const int T = n_threads(len); std::vector<uint64_t> c(T, 0);
    parallel_ranges(len, T, [&](uint64_t a, uint64_t b, int t) { uint64_t k = 0;
        for (uint64_t i = a; i < b; ++i) { const uint8_t ch = seq[i]; k += !(ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T' || ch == 'N'); } c[t] = k; });
    uint64_t tot = 0; for (auto v : c) tot += v; *count = tot;
    return NGSID_OK;
}

// repr(float) of CPython 3 for every v[i] (the '_score' suffix of sorted.fastq, get_sorted_fastq_for_cluster.py:176): the shortest digit string
// that round-trips (std::to_chars finds the same digits as CPython's dtoa mode 0), laid out by CPython's rule - fixed notation with at least
// ".0" when -4 <= decimal exponent < 16, else d.ddde+XX with an exponent of at least two digits; inf / nan as Python prints them.
// Each string is preceded by `prefix` when prefix != 0.  off[n+1] receives the CSR offsets; returns NGSID_ERR_CAPACITY (and *needed) when
// buf is too small (32 bytes per value always suffice).
extern "C" int32_t ngsid_host_repr_doubles(const double* v, uint64_t n, int32_t prefix, uint8_t* buf, uint64_t cap, uint64_t* off, uint64_t* needed)
{
    if ((n && (!v || !off)) || (!buf && cap)) return NGSID_ERR_ARG;
    if (needed) *needed = n * 32;
    if (cap < n * 32) return NGSID_ERR_CAPACITY;
    std::vector<uint8_t> len(n);
    parallel_ranges(n, n_threads(n * 64), [&](uint64_t a, uint64_t b, int) { for (uint64_t i = a; i < b; ++i) {
        char* o = (char*)buf + i * 32; char* p = o; const double x = v[i];
        if (prefix) *p++ = (char)prefix;
        if (std::isnan(x)) { memcpy(p, "nan", 3); p += 3; }
        else if (std::isinf(x)) { if (x < 0) *p++ = '-'; memcpy(p, "inf", 3); p += 3; }
        else {
            char sci[48]; auto r = std::to_chars(sci, sci + 40, x, std::chars_format::scientific); *r.ptr = 0;      // [-]d[.ddd]e[+-]XX, shortest round-trip digits
            const char* q = sci; if (*q == '-') { *p++ = '-'; ++q; }
            char dig[24]; int nd = 0; const char* e = q; while (e < r.ptr && *e != 'e') { if (*e != '.') dig[nd++] = *e; ++e; }
            const int ex = atoi(e + 1);                        // decimal exponent of the first digit
            if (ex >= -4 && ex < 16) {
                if (ex >= 0) { for (int k = 0; k <= ex; ++k) *p++ = k < nd ? dig[k] : '0'; *p++ = '.'; if (nd > ex + 1) { for (int k = ex + 1; k < nd; ++k) *p++ = dig[k]; } else *p++ = '0'; }
                else { *p++ = '0'; *p++ = '.'; for (int k = 0; k < -ex - 1; ++k) *p++ = '0'; for (int k = 0; k < nd; ++k) *p++ = dig[k]; }
            } else {
                *p++ = dig[0]; if (nd > 1) { *p++ = '.'; for (int k = 1; k < nd; ++k) *p++ = dig[k]; }
                *p++ = 'e'; *p++ = ex < 0 ? '-' : '+'; const int ae = ex < 0 ? -ex : ex; char t[8]; int nt = 0; int z = ae; do { t[nt++] = (char)('0' + z % 10); z /= 10; } while (z);
                if (nt < 2) t[nt++] = '0'; while (nt) *p++ = t[--nt];
            }
        }
        len[i] = (uint8_t)(p - o); } });
    // compact the 32-byte slots
    off[0] = 0; for (uint64_t i = 0; i < n; ++i) off[i + 1] = off[i] + len[i];
    for (uint64_t i = 0; i < n; ++i) if (off[i] != i * 32) memmove(buf + off[i], buf + i * 32, len[i]);
    return NGSID_OK;
}

// (f4) Infix ("HW") edit-distance location of a primer in a consensus end - what barcode_trimmer.find_barcode_locations gets from
// edlib.align(primer, center_window, mode="HW", task="locations", k=max_ed, additionalEqualities=IUPAC_map) (barcode_trimmer.py:34-60):
// unit costs, target ends free; *ed = the smallest edit distance (-1 when it exceeds max_ed), *end = the first (smallest) target position at
// which an alignment with that distance ends (inclusive), *start = the smallest start of such an alignment ending there (edlib takes the LAST
// end of the reversed prefix alignment for exactly that reason) - i.e. result["locations"][0].  iupac != 0 adds the reference's equalities
// (M = A/C, R = A/G, ... N, X = any of ACGT; symmetric, not transitive), case-sensitive like edlib.
// Bit-parallel (Myers 1999 / Hyyro) over 64-row blocks with per-character match masks; primers of any length.
namespace {
bool iupac_eq(uint8_t a, uint8_t b, int iupac)
{
    if (a == b) return true;
    if (!iupac) return false;
    auto set = [](uint8_t c) -> const char* { switch (c) { case 'M': return "AC"; case 'R': return "AG"; case 'W': return "AT"; case 'S': return "CG"; case 'Y': return "CT"; case 'K': return "GT";
        case 'V': return "ACG"; case 'H': return "ACT"; case 'D': return "AGT"; case 'B': return "CGT"; case 'X': return "ACGT"; case 'N': return "ACGT"; default: return ""; } };
    for (const char* p = set(a); *p; ++p) if ((uint8_t)*p == b) return true;
    for (const char* p = set(b); *p; ++p) if ((uint8_t)*p == a) return true;
    return false;
}
// last-row scores D[qlen][j] for j = 1..tlen of the alignment of q against t with a free start in t (prefix_mode == 0) or anchored at t[0] (1)
void myers_last_row(const uint8_t* q, int qlen, const uint8_t* t, int tlen, int iupac, int prefix_mode, std::vector<int>& last)
{
    const int nb = (qlen + 63) / 64;
    std::vector<uint64_t> peq((size_t)256 * nb, 0);
    bool seen[256] = {false};
    for (int j = 0; j < tlen; ++j) if (!seen[t[j]]) { seen[t[j]] = true; for (int i = 0; i < qlen; ++i) if (iupac_eq(q[i], t[j], iupac)) peq[(size_t)t[j] * nb + i / 64] |= 1ull << (i & 63); }
    std::vector<uint64_t> Pv(nb, ~0ull), Mv(nb, 0);
    std::vector<int> score(nb);                                  // score at the last row of every block
    for (int b = 0; b < nb; ++b) score[b] = std::min(qlen, (b + 1) * 64);
    last.assign(tlen, 0);
    const int lastbits = qlen - (nb - 1) * 64;                   // rows in the last block
    for (int j = 0; j < tlen; ++j) {
        int hin = prefix_mode ? 1 : 0;                           // horizontal delta entering the top row: 0 = free start in the target, +1 = anchored
        for (int b = 0; b < nb; ++b) {
            const uint64_t Eq = peq[(size_t)t[j] * nb + b];
            uint64_t pv = Pv[b], mv = Mv[b];
            const uint64_t hinNeg = hin < 0 ? 1ull : 0ull;
            uint64_t Xv = Eq | mv;
            const uint64_t Eq2 = Eq | hinNeg;
            uint64_t Xh = (((Eq2 & pv) + pv) ^ pv) | Eq2;
            uint64_t Ph = mv | ~(Xh | pv);
            uint64_t Mh = pv & Xh;
            const int top = (b == nb - 1) ? lastbits - 1 : 63;
            int hout = 0; if ((Ph >> top) & 1) hout = 1; else if ((Mh >> top) & 1) hout = -1;
            Ph <<= 1; Mh <<= 1;
            if (hin < 0) Mh |= 1ull; else if (hin > 0) Ph |= 1ull;
            Pv[b] = Mh | ~(Xv | Ph); Mv[b] = Ph & Xv;
            score[b] += hout; hin = hout;
        }
        last[j] = score[nb - 1];
    }
}
}  // namespace

extern "C" int32_t ngsid_host_infix_locate(const uint8_t* query, int32_t qlen, const uint8_t* target, int32_t tlen, int32_t max_ed, int32_t iupac,
                                           int32_t* ed, int32_t* start, int32_t* end)
{
    if (!ed || !start || !end || (qlen > 0 && !query) || (tlen > 0 && !target) || qlen < 0 || tlen < 0) return NGSID_ERR_ARG;
    *ed = -1; *start = -1; *end = -1;
    if (qlen == 0 || tlen == 0) return NGSID_OK;
    std::vector<int> last;
    myers_last_row(query, qlen, target, tlen, iupac, 0, last);
    int best = qlen;                                           // the empty alignment (all of the primer deleted ...) is not a location: edlib needs an end position
    int e = -1;
    for (int j = 0; j < tlen; ++j) if (last[j] < best) { best = last[j]; e = j; }
    if (e < 0 || (max_ed >= 0 && best > max_ed)) return NGSID_OK;
    // start: reversed primer against the reversed target prefix, anchored at the end position: the LAST position with the same distance
    std::vector<uint8_t> rq(query, query + qlen), rt(target, target + e + 1);
    std::reverse(rq.begin(), rq.end()); std::reverse(rt.begin(), rt.end());
    std::vector<int> rl; myers_last_row(rq.data(), qlen, rt.data(), e + 1, iupac, 1, rl);
    int jl = -1; for (int j = 0; j <= e; ++j) if (rl[j] == best) jl = j;
    *ed = best; *end = e; *start = jl >= 0 ? e - jl : e + 1;     // jl < 0: the alignment consumes no target base before `end` ... cannot happen with best < qlen
    return NGSID_OK;
}
