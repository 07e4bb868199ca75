// k_cluster.hip - (a4-a11) greedy clustering: representative index, hit counting, mapping criterion,
// alignment criterion, and the speculative-block driver that reproduces the sequential semantics of
// cluster.reads_to_clusters (cluster.py:207-353) exactly.
//
// Data in HBM
//   reads            CSR (bases, qualities, uint64 offsets), score order = greedy order
//   minimizers       (code u64, pos u32) written sparsely at the read's base offset by k_hpc_minimizers
//   representatives  slot -> read id; per slot the SORTED UNIQUE codes of its minimizers (pool + offsets)
//   index "DB"       all (code, slot) pairs sorted by code = minimizer_database (cluster.py:329-334)
//   hit matrix       block_items x stride uint64, each cell = n_hits<<48 | sum(pos)  (get_all_hits :43-62)
// Exactness of the parallel form: a block of reads is evaluated against a database snapshot; the first
// read (in greedy order) that founds a new cluster is committed together with everything before it, its
// minimizers are merged into the index, only the DELTA (hits against that one representative) is added to
// the hit matrix of the later reads, and those are re-decided.  Aligner results are cached per
// (read, representative) since they do not depend on the database.
#include "ngsid_internal.h"
#include "../../include/ngsid_tables.h"
#include <math.h>
#include <algorithm>

#define DEC_NEWREP   (-1)
#define DEC_SHORT    (-2)
#define DEC_PENDING  (-3)
#define DEC_UNDEC    (-4)
#define NCACHE 4

__constant__ double c_round2_t[15];
static bool g_cl_tables[16] = {false};

struct ClDev {
    const uint8_t* seq; const uint8_t* qual; const uint64_t* off; uint64_t n;
    const uint32_t* hlen; const uint32_t* mzcnt; const double* herr; const double* rawerr; const uint8_t* eidx; const uint32_t* accrank;
    const uint64_t* mzcode; const uint32_t* mzpos; const uint64_t* mzoff;      // compact CSR: the minimizers of read r start at mzoff[r] (round 5; round 1-4: at the read's base offset)
    const uint32_t* rep_read; const uint64_t* pool; const uint64_t* pool_off;
    const int32_t* maxgap;      // 225 ints, -2 = missing table entry
    int k, min_shared, symmetric; double min_fraction, mapped_threshold, aligned_threshold;
    // per read state
    int32_t* dec; uint8_t* kind; uint8_t* alnflag; int32_t* top;
    uint64_t* cur_hi; uint64_t* cur_lo;
    int32_t* cache_slot; int32_t* cache_region; uint8_t* cache_ptr;
    int* errflag;
};

__global__ void k_eidx(const double* __restrict__ herr, const double* __restrict__ known, uint64_t n, double* __restrict__ herr_out, uint8_t* __restrict__ eidx)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double e = herr[i];
    if (known) { const double kv = known[i]; if (kv == kv) e = kv; }
    herr_out[i] = e;
    int j = 0;
    while (j < 15 && e >= c_round2_t[j]) ++j;       // round(e,2) clamped to [0.01,0.15]  cluster.py:356-366
    if (j < 1) j = 1;
    eidx[i] = (uint8_t)j;
}

// ---- hit counting: one wave per item, lanes over the read's minimizers, binary search in the sorted index
__global__ __launch_bounds__(256)
void k_count_hits(ClDev D, const uint32_t* __restrict__ items, uint32_t it_lo, uint32_t it_hi, uint32_t row0,
                  const uint64_t* __restrict__ db_code, const uint32_t* __restrict__ db_slot, uint32_t const_slot, uint64_t n_db,
                  uint64_t* __restrict__ cnt, uint32_t stride)
{
    // db_slot == nullptr: the index is one representative's own sorted unique codes (delta pass), slot = const_slot
    const int lane = threadIdx.x & 63;
    const uint32_t it = it_lo + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (it >= it_hi || n_db == 0) return;
    const uint32_t read = items[it];
    if (D.hlen[read] < (uint32_t)D.k) return;
    const uint32_t M = D.mzcnt[read];
    const uint64_t base = D.mzoff[read];
    unsigned long long* row = (unsigned long long*)(cnt + (uint64_t)(it - row0) * stride);
    for (uint32_t a = lane; a < M; a += 64) {
        const uint64_t code = D.mzcode[base + a]; const uint32_t pos = D.mzpos[base + a];
        uint64_t lo = 0, hi = n_db;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (db_code[mid] < code) lo = mid + 1; else hi = mid; }
        for (uint64_t p = lo; p < n_db && db_code[p] == code; ++p)
            atomicAdd(&row[db_slot ? db_slot[p] : const_slot], (1ull << 48) + (unsigned long long)pos);
    }
}

// wave-wide maximum of a 128-bit key (hi,lo); every lane returns the winner
__device__ __forceinline__ void wave_max128(uint64_t& hi, uint64_t& lo)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint64_t oh = __shfl_xor((unsigned long long)hi, d), ol = __shfl_xor((unsigned long long)lo, d);
        if (oh > hi || (oh == hi && ol > lo)) { hi = oh; lo = ol; }
    }
}

// candidate order = sorted(key=(len(hits), sum(pos), accession), reverse=True)  cluster.py:79,174 ; full ties: lower slot first
__device__ __forceinline__ bool next_candidate(const ClDev& D, const uint64_t* row, uint32_t R, int lane, uint64_t bound_hi, uint64_t bound_lo,
                                               int need_n /* -1 = any */, uint64_t& out_hi, uint64_t& out_lo)
{
    uint64_t bh = 0, bl = 0;
    for (uint32_t s = lane; s < R; s += 64) {
        const uint64_t v = row[s];
        if ((v >> 48) == 0) continue;
        if (need_n >= 0 && (int)(v >> 48) != need_n) continue;
        const uint64_t lo = ((uint64_t)D.accrank[D.rep_read[s]] << 32) | (uint64_t)(0xffffffffu - s);
        if (v > bound_hi || (v == bound_hi && lo >= bound_lo)) continue;        // strictly below the bound
        if (v > bh || (v == bh && lo > bl)) { bh = v; bl = lo; }
    }
    wave_max128(bh, bl);
    out_hi = bh; out_lo = bl;
    return bh != 0;
}

// mapped span of a read against one representative (the body of get_best_cluster, cluster.py:92-117):
// hit = read minimizer present in the representative's code set; a gap of g non-hit minimizers between two hits counts as
// mapped iff p_err^g >= min_prob_no_hits <=> g <= maxgap.
__device__ __forceinline__ long long mapped_span(const ClDev& D, uint32_t read, uint32_t slot, int maxgap, int lane)
{
    const uint32_t M = D.mzcnt[read]; const uint64_t base = D.mzoff[read];
    const uint64_t* rc = D.pool + D.pool_off[slot]; const uint32_t rn = (uint32_t)(D.pool_off[slot + 1] - D.pool_off[slot]);
    long long total = 0;
    int last_idx = -1, last_pos = 0;
    for (uint32_t c0 = 0; c0 < M; c0 += 64) {
        const uint32_t a = c0 + lane;
        bool hit = false; int pos = 0;
        if (a < M) {
            const uint64_t code = D.mzcode[base + a]; pos = (int)D.mzpos[base + a];
            uint32_t lo = 0, hi = rn;
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rc[mid] < code) lo = mid + 1; else hi = mid; }
            hit = lo < rn && rc[lo] == code;
        }
        const unsigned long long mask = __ballot(hit);
        const unsigned long long below = mask & ((lane == 0) ? 0ull : (~0ull >> (64 - lane)));
        const int pl = below ? 63 - __clzll(below) : 0;
        const int ppos = __shfl(pos, pl);                       // unconditional: no cross-lane op under divergence
        const int prev_idx = below ? (int)c0 + pl : last_idx;
        const int prev_pos = below ? ppos : last_pos;
        long long contrib = 0;
        if (hit) { const int gap = (int)a - prev_idx - 1; if (gap <= maxgap) contrib = pos - prev_pos; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) contrib += __shfl_xor(contrib, d);
        total += contrib;
        if (mask) { const int hl = 63 - __clzll(mask); last_idx = (int)c0 + hl; last_pos = __shfl(pos, hl); }
    }
    if (last_idx >= 0) { const int gap = (int)M - 1 - last_idx; if (gap <= maxgap) total += (long long)D.hlen[read] - last_pos; }
    return total;
}

// ---- mapping criterion for every undecided item of the block (get_best_cluster, cluster.py:67-127)
__global__ __launch_bounds__(256)
void k_decide_map(ClDev D, const uint32_t* __restrict__ items, uint32_t it_lo, uint32_t it_hi, uint32_t row0,
                  const uint64_t* __restrict__ cnt, uint32_t stride, uint32_t R)
{
    const int lane = threadIdx.x & 63;
    const uint32_t it = it_lo + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (it >= it_hi) return;
    const uint32_t read = items[it];
    if (D.dec[read] != DEC_UNDEC) return;                 // decided in an earlier pass of this block and not affected by the representatives committed since (k_reset_items)
    if (D.hlen[read] < (uint32_t)D.k) { if (lane == 0) { D.dec[read] = DEC_SHORT; D.alnflag[read] = 0; } return; }
    const uint64_t* row = cnt + (uint64_t)(it - row0) * stride;
    int top = 0;
    for (uint32_t s = lane; s < R; s += 64) top = max(top, (int)(row[s] >> 48));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) top = max(top, __shfl_xor(top, d));
    if (lane == 0) { D.top[read] = top; D.alnflag[read] = 0; D.cur_hi[read] = ~0ull; D.cur_lo[read] = ~0ull; }
    if (top < D.min_shared) { if (lane == 0) D.dec[read] = DEC_NEWREP; return; }         // :82-83, and :310 fails
    uint64_t bh = ~0ull, bl = ~0ull;
    const int i1 = D.eidx[read];
    const double hl = (double)D.hlen[read];
    for (;;) {
        uint64_t kh, kl;
        if (!next_candidate(D, row, R, lane, bh, bl, -1, kh, kl)) break;
        const int nm = (int)(kh >> 48);
        if ((double)nm < D.min_fraction * (double)top || nm < D.min_shared) break;          // :88
        const uint32_t slot = 0xffffffffu - (uint32_t)(kl & 0xffffffffu);
        const uint32_t rr = D.rep_read[slot];
        const int mg = D.maxgap[(i1 - 1) * 15 + (D.eidx[rr] - 1)];
        if (mg == -2) { if (lane == 0) atomicExch(D.errflag, 1); break; }
        const long long tm = mapped_span(D, read, slot, mg, lane);
        const double ratio = (double)tm / hl;                                               // :117
        bool pass;
        if (D.symmetric) { const double rratio = (double)tm / (double)D.hlen[rr]; pass = fmin(ratio, rratio) > D.mapped_threshold; }
        else pass = ratio > D.mapped_threshold;
        if (pass) { if (lane == 0) { D.dec[read] = (int32_t)slot; D.kind[read] = NGSID_ST_MAPPED; } return; }
        bh = kh; bl = kl;
    }
    if (lane == 0) { D.dec[read] = DEC_PENDING; D.alnflag[read] = 1; }                      // :310-311
}

// ---- alignment criterion cursor (get_best_cluster_block_align, cluster.py:172-205): resolve from cache or request a pair
__global__ __launch_bounds__(1024)
void k_aln_next(ClDev D, const uint32_t* __restrict__ items, uint32_t it_lo, uint32_t it_hi, uint32_t row0,
                const uint64_t* __restrict__ cnt, uint32_t stride, uint32_t R,
                uint32_t* __restrict__ req_q, uint32_t* __restrict__ req_t, uint32_t* __restrict__ req_slot, int32_t* __restrict__ req_open, int32_t* __restrict__ req_mid,
                uint32_t* __restrict__ req_count)
{
    // one wave per read, 16 reads per workgroup; the requests of a workgroup take their list slots with ONE atomic (a million single
    // atomics on the same counter would serialise)
    __shared__ uint32_t s_want[16]; __shared__ uint32_t s_base;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t it = it_lo + blockIdx.x * (blockDim.x >> 6) + wv;
    bool want = false; uint32_t q_read = 0, q_rr = 0, q_slot = 0; int q_open = 0, q_mid = 0;
    if (it < it_hi) {
        const uint32_t read = items[it];
        if (D.dec[read] == DEC_PENDING) {
            const uint64_t* row = cnt + (uint64_t)(it - row0) * stride;
            const int top = D.top[read];
            uint64_t bh = D.cur_hi[read], bl = D.cur_lo[read];
            const double qlen = (double)(D.off[read + 1] - D.off[read]);
            for (;;) {
                uint64_t kh, kl;
                if (!next_candidate(D, row, R, lane, bh, bl, top, kh, kl)) { if (lane == 0) D.dec[read] = DEC_NEWREP; break; }   // :181 / :205
                const uint32_t slot = 0xffffffffu - (uint32_t)(kl & 0xffffffffu);
                const uint32_t rr = D.rep_read[slot];
                int region = -1;
                for (int c = 0; c < NCACHE; ++c) if (D.cache_slot[(uint64_t)read * NCACHE + c] == (int32_t)slot) region = D.cache_region[(uint64_t)read * NCACHE + c];
                if (region < 0) {
                    const double ers = D.rawerr[read] + D.rawerr[rr];                                         // :188
                    q_open = ers <= 0.01 ? 5 : (ers <= 0.04 ? 4 : (ers <= 0.1 ? 3 : 2));                      // :189-196
                    q_mid = (int)floor((1.0 - ers) * (double)D.k);                                            // :198
                    q_read = read; q_rr = rr; q_slot = slot; want = true;
                    if (lane == 0) { D.cur_hi[read] = bh; D.cur_lo[read] = bl; }
                    break;
                }
                const double ar = (double)region / qlen;                                                          // :167
                bool pass;
                if (D.symmetric) { const double tr = (double)region / (double)(D.off[rr + 1] - D.off[rr]); pass = fmin(ar, tr) >= D.aligned_threshold; }
                else pass = ar >= D.aligned_threshold;
                if (pass) { if (lane == 0) { D.dec[read] = (int32_t)slot; D.kind[read] = NGSID_ST_ALIGNED; } break; }
                bh = kh; bl = kl;
            }
        }
    }
    if (lane == 0) s_want[wv] = want ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { const uint32_t w = s_want[i]; s_want[i] = n; n += w; }
        s_base = n ? atomicAdd(req_count, n) : 0u;
    }
    __syncthreads();
    if (want && lane == 0) {
        const uint32_t idx = s_base + s_want[wv];
        req_q[idx] = q_read; req_t[idx] = q_rr; req_slot[idx] = q_slot; req_open[idx] = q_open; req_mid[idx] = q_mid;
    }
}

__global__ void k_cache_insert(ClDev D, const uint32_t* __restrict__ req_q, const uint32_t* __restrict__ req_slot, const int32_t* __restrict__ region, uint32_t nreq)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nreq) return;
    const uint32_t read = req_q[i];                 // one outstanding request per read -> no race on its cache line
    const uint8_t ptr = D.cache_ptr[read];
    D.cache_slot[(uint64_t)read * NCACHE + (ptr % NCACHE)] = (int32_t)req_slot[i];
    D.cache_region[(uint64_t)read * NCACHE + (ptr % NCACHE)] = region[i];
    D.cache_ptr[read] = (uint8_t)((ptr + 1) % NCACHE);
}

// one bit per block item: the item decided "new representative" (word w covers the items row0 + 64 w ... of the block)
// Round 5: the word behind the mask (index w_tail) carries the round's two scalars - pairs requested by k_aln_next (low half) and the p-table error flag (high half) - so that a
// round's results travel to the host as ONE copy instead of three, and the request counter is reset here, by its reader, instead of by a memset in front of the next k_aln_next.
__global__ void k_newrep_mask(const int32_t* __restrict__ dec, const uint32_t* __restrict__ items, uint32_t it_first /* row0 + multiple of 64 */, uint32_t it_lo, uint32_t it_hi,
                              uint32_t row0, unsigned long long* __restrict__ mask, uint32_t w_tail, uint32_t* __restrict__ nreq, const int* __restrict__ eflag)
{
    const uint32_t it = it_first + blockIdx.x * blockDim.x + threadIdx.x;
    const bool v = it >= it_lo && it < it_hi && dec[items[it]] == DEC_NEWREP;
    const unsigned long long bits = __ballot(v);
    if ((threadIdx.x & 63) == 0 && it < it_hi) mask[(it - row0) >> 6] = bits;
    if (blockIdx.x == 0 && threadIdx.x == 0) { mask[w_tail] = ((unsigned long long)(unsigned)*eflag << 32) | (unsigned long long)*nreq; *nreq = 0u; }
}

// hits of the block items [it_lo,it_hi) against ONE freshly built representative (its sorted unique codes in the pool; the offsets are read on the
// device, so tentative representatives need no host round trip)
__global__ __launch_bounds__(256)
void k_count_hits_rep(ClDev D, const uint32_t* __restrict__ items, uint32_t it_lo, uint32_t it_hi, uint32_t row0, uint32_t slot, uint64_t* __restrict__ cnt, uint32_t stride)
{
    const int lane = threadIdx.x & 63;
    const uint32_t it = it_lo + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (it >= it_hi) return;
    const uint32_t read = items[it];
    if (D.hlen[read] < (uint32_t)D.k) return;
    const uint64_t* __restrict__ rc = D.pool + D.pool_off[slot];
    const uint32_t n = (uint32_t)(D.pool_off[slot + 1] - D.pool_off[slot]);
    const uint32_t M = D.mzcnt[read];
    const uint64_t base = D.mzoff[read];
    unsigned long long* cell = (unsigned long long*)(cnt + (uint64_t)(it - row0) * stride + slot);
    unsigned long long acc = 0;
    for (uint32_t a = lane; a < M; a += 64) {
        const uint64_t code = D.mzcode[base + a];
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rc[mid] < code) lo = mid + 1; else hi = mid; }
        if (lo < n && rc[lo] == code) acc += (1ull << 48) + (unsigned long long)D.mzpos[base + a];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0 && acc) *cell += acc;
}

// Can a representative that shares nm minimizers with `read` change the decision the read took WITHOUT it?  Only if the reference would ever look at it: get_best_cluster walks
// the candidates by hit count and stops at the first one below min_shared or below min_fraction x top (cluster.py:82,88), the alignment stage only takes candidates AT the top count
// (:181) - so a representative below that bound is never visited and, being below the top, does not move the top either.  D.top[read] is the top count over the committed
// representatives as of the read's last decision; it stays valid while the read is unaffected, because a representative that could raise it is one that "matters" here.
// (Round 6.  Until round 5 the test was nm >= min_shared alone: in a noisy set nearly every read shares five minimizers with every new noise representative of its species, so
// almost every restart round committed a single representative.)
__device__ __forceinline__ bool rep_can_matter(const ClDev& D, uint32_t read, int nm)
{
    // (min_fraction above 1 - a legal flag value - makes the walk stop at once, but a representative AT or above the top count still changes the top / joins the alignment candidates)
    const double mf = D.min_fraction < 1.0 ? D.min_fraction : 1.0;
    return nm >= D.min_shared && !((double)nm < mf * (double)D.top[read]);
}

// first block item in [it_lo,it_hi) that shares at least min_shared minimizers with one of the T tentative representatives in the columns
// R0 .. R0+T-1 (only such an item can decide differently once they exist: get_best_cluster breaks below min_shared, cluster.py:82,88, and the
// alignment stage only looks at the top count, :181, which is >= min_shared, :310)
__global__ __launch_bounds__(256)
void k_first_affected(ClDev D, const uint32_t* __restrict__ items, uint32_t it_lo, uint32_t it_hi, uint32_t row0, const uint64_t* __restrict__ cnt, uint32_t stride,
                      uint32_t R0, uint32_t T, uint32_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const uint32_t it = it_lo + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (it >= it_hi) return;
    const uint64_t* row = cnt + (uint64_t)(it - row0) * stride;
    const int nm = (uint32_t)lane < T ? (int)(row[R0 + lane] >> 48) : 0;
    const bool hit = rep_can_matter(D, items[it], nm);
    if (__ballot(hit) != 0ull && lane == 0) atomicMin(out, it);
}

// sort + unique the minimizer codes of one read into the representative pool (single workgroup, bitonic in LDS);
// also registers the slot: rep_read[slot] = read, pool_off[slot+1] = pool_off[slot] + #unique
__global__ __launch_bounds__(256)
void k_rep_build(const uint64_t* __restrict__ mzcode, uint64_t base, uint32_t M, uint32_t P2, uint64_t* __restrict__ pool, uint64_t* __restrict__ pool_off,
                 uint32_t* __restrict__ rep_read, uint32_t slot, uint32_t read, uint32_t* __restrict__ out_count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* a = (uint64_t*)smem;
    for (uint32_t i = threadIdx.x; i < P2; i += 256) a[i] = i < M ? mzcode[base + i] : ~0ull;
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1)
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P2; i += 256) {
                const uint32_t l = i ^ j;
                if (l > i) { const uint64_t x = a[i], y = a[l]; const bool up = (i & k2) == 0; if ((x > y) == up) { a[i] = y; a[l] = x; } }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        uint64_t* dst = pool + pool_off[slot];
        uint32_t c = 0; for (uint32_t i = 0; i < M; ++i) if (i == 0 || a[i] != a[i - 1]) dst[c++] = a[i];
        pool_off[slot + 1] = pool_off[slot] + c; rep_read[slot] = read; *out_count = c;
    }
}

__global__ void k_db_merge(const uint64_t* __restrict__ oc, const uint32_t* __restrict__ os, uint64_t n_old,
                           const uint64_t* __restrict__ rc, uint32_t n_new, uint32_t slot, uint64_t* __restrict__ nc, uint32_t* __restrict__ ns)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_old) {
        const uint64_t code = oc[i]; uint32_t lo = 0, hi = n_new;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rc[mid] < code) lo = mid + 1; else hi = mid; }   // rep codes < old code
        nc[i + lo] = code; ns[i + lo] = os[i];
    } else if (i < n_old + n_new) {
        const uint32_t j = (uint32_t)(i - n_old); const uint64_t code = rc[j]; uint64_t lo = 0, hi = n_old;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (oc[mid] <= code) lo = mid + 1; else hi = mid; }  // old codes <= rep code
        nc[j + lo] = code; ns[j + lo] = slot;
    }
}

// ---- batched variants (up to 64 representatives per launch): one launch builds them, one counts the block against them, one merges them
struct RepBatch { uint32_t read[64]; uint32_t M[64]; uint64_t base[64]; };
struct PosBatch { uint32_t pos[64]; };

// workgroup t sorts + uniques the codes of read t (bitonic in LDS) and appends them to the pool behind workgroup t-1: the offsets are chained
// through `chain` (zero before the launch; all <= 64 workgroups are resident, so waiting on the predecessor cannot deadlock)
__global__ __launch_bounds__(256)
void k_rep_build_batch(const uint64_t* __restrict__ mzcode, RepBatch B, uint64_t* __restrict__ pool, uint64_t* __restrict__ pool_off,
                       uint32_t* __restrict__ rep_read, uint32_t slot0, uint32_t* __restrict__ chain)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* a = (uint64_t*)smem;
    __shared__ uint32_t s_u; __shared__ uint64_t s_off;
    const uint32_t t = blockIdx.x, M = B.M[t]; const uint64_t base = B.base[t];
    uint32_t P2 = 2; while (P2 < M) P2 <<= 1;
    for (uint32_t i = threadIdx.x; i < P2; i += 256) a[i] = i < M ? mzcode[base + i] : ~0ull;
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= P2; k2 <<= 1)
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P2; i += 256) {
                const uint32_t l = i ^ j;
                if (l > i) { const uint64_t x = a[i], y = a[l]; const bool up = (i & k2) == 0; if ((x > y) == up) { a[i] = y; a[l] = x; } }
            }
            __syncthreads();
        }
    if (threadIdx.x == 0) {
        uint32_t c = 0; for (uint32_t i = 0; i < M; ++i) if (i == 0 || a[i] != a[i - 1]) a[c++] = a[i];      // in place: c <= i
        while (__atomic_load_n(chain, __ATOMIC_ACQUIRE) != t) __builtin_amdgcn_s_sleep(1);
        const uint64_t off = pool_off[slot0 + t];
        pool_off[slot0 + t + 1] = off + c; rep_read[slot0 + t] = B.read[t];
        __threadfence();
        __atomic_store_n(chain, t + 1, __ATOMIC_RELEASE);
        s_u = c; s_off = off;
    }
    __syncthreads();
    const uint32_t u = s_u; uint64_t* dst = pool + s_off;
    for (uint32_t i = threadIdx.x; i < u; i += 256) dst[i] = a[i];
}

// hits of the block items behind position P.pos[t] against the tentative representative in slot slot0 + t (blockIdx.y = t)
__global__ __launch_bounds__(256)
void k_count_hits_reps(ClDev D, const uint32_t* __restrict__ items, uint32_t it_lo, uint32_t it_hi, uint32_t row0, uint32_t slot0, PosBatch P,
                       uint64_t* __restrict__ cnt, uint32_t stride)
{
    const int lane = threadIdx.x & 63;
    const uint32_t it = it_lo + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t t = blockIdx.y, slot = slot0 + t;
    if (it >= it_hi || it <= P.pos[t]) return;
    const uint32_t read = items[it];
    if (D.hlen[read] < (uint32_t)D.k) return;
    const uint64_t* __restrict__ rc = D.pool + D.pool_off[slot];
    const uint32_t n = (uint32_t)(D.pool_off[slot + 1] - D.pool_off[slot]);
    const uint32_t M = D.mzcnt[read];
    const uint64_t base = D.mzoff[read];
    unsigned long long acc = 0;
    for (uint32_t a = lane; a < M; a += 64) {
        const uint64_t code = D.mzcode[base + a];
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (rc[mid] < code) lo = mid + 1; else hi = mid; }
        if (lo < n && rc[lo] == code) acc += (1ull << 48) + (unsigned long long)D.mzpos[base + a];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0 && acc) cnt[(uint64_t)(it - row0) * stride + slot] += acc;
}

// merges the sorted code lists of the c consecutive slots slot0 .. slot0+c-1 (contiguous in the pool) into the sorted index in one pass; the
// result equals c successive k_db_merge calls: equal codes keep the order old entries, then ascending slot
__global__ void k_db_merge_batch(const uint64_t* __restrict__ oc, const uint32_t* __restrict__ os, uint64_t n_old,
                                 const uint64_t* __restrict__ pool, const uint64_t* __restrict__ pool_off, uint32_t slot0, uint32_t c,
                                 uint64_t* __restrict__ nc, uint32_t* __restrict__ ns)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = pool_off[slot0], n_new = pool_off[slot0 + c] - p0;
    if (i >= n_old + n_new) return;
    uint64_t code, idx; uint32_t slot, mine = 0xffffffffu;
    if (i < n_old) { code = oc[i]; slot = os[i]; idx = i; }
    else {
        const uint64_t j = p0 + (i - n_old);                         // position in the pool: which list?
        uint32_t lo = 0, hi = c; while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pool_off[slot0 + mid] <= j) lo = mid; else hi = mid; }
        mine = lo; code = pool[j]; slot = slot0 + lo; idx = j - pool_off[slot0 + lo];
        uint64_t l2 = 0, h2 = n_old; while (l2 < h2) { const uint64_t mid = (l2 + h2) >> 1; if (oc[mid] <= code) l2 = mid + 1; else h2 = mid; }   // old codes <= code
        idx += l2;
    }
    for (uint32_t t = 0; t < c; ++t) {
        if (t == mine) continue;
        const uint64_t* l = pool + pool_off[slot0 + t]; const uint32_t n = (uint32_t)(pool_off[slot0 + t + 1] - pool_off[slot0 + t]);
        const bool incl = mine != 0xffffffffu && t < mine;           // lists before mine: their codes <= code come first; old entries and later lists: only codes < code
        uint32_t lo = 0, hi = n;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (incl ? l[mid] <= code : l[mid] < code) lo = mid + 1; else hi = mid; }
        idx += lo;
    }
    nc[idx] = code; ns[idx] = slot;
}

// After `nnew` tentative representatives (hit-matrix columns R0 .. R0 + nnew) were committed, only the items that share >= min_shared minimizers
// with one of them are decided again.  Every other item keeps its decision: a representative below min_shared hits is never a candidate of the
// mapping stage (cluster.py:82,88), never ties the top count of the alignment stage (:181, top >= min_shared) and does not change the top count,
// so the walk over the candidates of such an item is the same as before - the argument behind k_first_affected, applied per item.  In noisy read
// sets (a new representative every ~200 reads) this takes the re-decided items per pass from "the rest of the block" to a few per cent of it.
__global__ void k_reset_items(ClDev D, const uint32_t* __restrict__ items, uint32_t it_lo, uint32_t it_hi, uint32_t row0, const uint64_t* __restrict__ cnt, uint32_t stride,
                              uint32_t R0, uint32_t nnew)
{
    const uint32_t it = it_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (it >= it_hi) return;
    const uint64_t* row = cnt + (uint64_t)(it - row0) * stride + R0;
    bool affected = false;
    const uint32_t read = items[it];
    for (uint32_t t = 0; t < nnew; ++t) affected = affected || rep_can_matter(D, read, (int)(row[t] >> 48));
    if (affected) D.dec[read] = DEC_UNDEC;
}

// round 6: items [it_lo, it_hi) leave the current block undecided (the driver cut the block short: a later block decides them from scratch)
__global__ void k_undecide(ClDev D, const uint32_t* __restrict__ items, uint32_t it_lo, uint32_t it_hi)
{
    const uint32_t it = it_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (it < it_hi) D.dec[items[it]] = DEC_UNDEC;
}

__global__ void k_finalize(ClDev D, uint64_t n, const uint8_t* __restrict__ seeded, int32_t* __restrict__ rep_of, uint8_t* __restrict__ status,
                           double* __restrict__ herr_out, unsigned long long* __restrict__ counters)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double nan = __longlong_as_double(0x7ff8000000000000ULL);
    int which = -1; bool aln = false;               // counter this read adds to (0 mapped, 1 aligned, 3 new representative); counted once per wave
    if (i < n) {
        const int32_t d = D.dec[i];
        if (seeded && seeded[i]) { rep_of[i] = (int32_t)i; status[i] = NGSID_ST_SEEDED; herr_out[i] = D.herr[i]; }
        else if (d == DEC_SHORT) { rep_of[i] = (int32_t)i; status[i] = NGSID_ST_SHORT; herr_out[i] = nan; }
        else {
            herr_out[i] = D.herr[i];
            aln = D.alnflag[i] != 0;
            if (d >= 0) { const uint8_t kd = D.kind[i]; rep_of[i] = (int32_t)D.rep_read[d]; status[i] = kd; which = kd == NGSID_ST_MAPPED ? 0 : 1; }
            else { rep_of[i] = (int32_t)i; status[i] = NGSID_ST_NEWREP; which = 3; }
        }
    }
    const unsigned long long c0 = __popcll(__ballot(which == 0)), c1 = __popcll(__ballot(which == 1)), c2 = __popcll(__ballot(aln)), c3 = __popcll(__ballot(which == 3));
    if ((threadIdx.x & 63) == 0) {
        if (c0) atomicAdd(&counters[0], c0);
        if (c1) atomicAdd(&counters[1], c1);
        if (c2) atomicAdd(&counters[2], c2);
        if (c3) atomicAdd(&counters[3], c3);
    }
}

// ------------------------------------------------------------------------------------------------ host driver
namespace {
struct RepStore {
    ngsid_ctx* ctx; uint32_t R = 0, Rcap = 0;
    DevBuf<uint32_t> rep_read; DevBuf<uint64_t> pool, pool_off; std::vector<uint64_t> h_pool_off;
    DevBuf<uint64_t> dbc[2]; DevBuf<uint32_t> dbs[2]; int cur = 0; uint64_t n_db = 0;
    DevBuf<uint32_t> d_count;
};
}

// Builds the representatives of `reads` in the slots S.R, S.R+1, ... (pool offsets chained on the device) without registering them on the host:
// the caller copies pool_off[S.R .. S.R+n] back and commits a prefix.  A slot that is not committed is simply overwritten by the next build.
// Builds the representatives of `reads` in the slots S.R, S.R+1, ... (pool offsets chained on the device) without registering them on the host:
// the caller copies pool_off[S.R .. S.R+n] back and commits a prefix.  A slot that is not committed is simply overwritten by the next build.
static int32_t build_reps(ngsid_ctx* ctx, RepStore& S, const uint32_t* reads, uint32_t n, const uint64_t* d_mzcode, const uint64_t* h_off, const uint32_t* h_mzcnt)
{
    if (S.R + n + 1 > S.Rcap) {
        uint32_t nc = S.Rcap ? S.Rcap : 1024; while (nc < S.R + n + 1) nc *= 2;
        HIPCHK(ctx, S.rep_read.grow(nc, ctx->stream)); HIPCHK(ctx, S.pool_off.grow((size_t)nc + 1, ctx->stream));
        S.Rcap = nc;
    }
    uint64_t need = S.h_pool_off[S.R];
    for (uint32_t t = 0; t < n; ++t) need += h_mzcnt[reads[t]];
    if (need + 1 > S.pool.n) HIPCHK(ctx, S.pool.grow(std::max<size_t>((need + 1) * 2, 1 << 16), ctx->stream));
    for (uint32_t t0 = 0; t0 < n; t0 += 64) {
        const uint32_t nb = std::min<uint32_t>(64, n - t0);
        RepBatch B; uint32_t maxM = 2;
        for (uint32_t t = 0; t < nb; ++t) { B.read[t] = reads[t0 + t]; B.M[t] = h_mzcnt[reads[t0 + t]]; B.base[t] = h_off[reads[t0 + t]]; maxM = std::max(maxM, B.M[t]); }
        uint32_t P2 = 2; while (P2 < maxM) P2 <<= 1;
        const size_t lds = (size_t)P2 * 8;
        if (lds > 48 * 1024) HIPCHK(ctx, hipFuncSetAttribute((const void*)k_rep_build_batch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(ctx, hipMemsetAsync(S.d_count.p, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_rep_build_batch, dim3(nb), dim3(256), lds, ctx->stream, d_mzcode, B, S.pool.p, S.pool_off.p, S.rep_read.p, S.R + t0, S.d_count.p);
        HIPCHK(ctx, hipGetLastError());
    }
    return NGSID_OK;
}

// Registers the first c built representatives: host mirror of the pool offsets (h_po = pool_off[S.R .. S.R+c] as read back) and the merged index.
static int32_t commit_reps(ngsid_ctx* ctx, RepStore& S, uint32_t c, const uint64_t* h_po)
{
    for (uint32_t t0 = 0; t0 < c; t0 += 64) {
        const uint32_t nb = std::min<uint32_t>(64, c - t0);
        const uint64_t n_new = h_po[t0 + nb] - h_po[t0];
        for (uint32_t t = 0; t < nb; ++t) S.h_pool_off.push_back(h_po[t0 + t + 1]);
        const int nx = S.cur ^ 1;
        if (S.n_db + n_new + 1 > S.dbc[nx].n) { const size_t cap = std::max<size_t>((S.n_db + n_new + 1) * 2, 1 << 16); HIPCHK(ctx, S.dbc[nx].alloc(cap)); HIPCHK(ctx, S.dbs[nx].alloc(cap)); }
        const uint64_t tot = S.n_db + n_new;
        if (tot) hipLaunchKernelGGL(k_db_merge_batch, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream,
                                    S.dbc[S.cur].p, S.dbs[S.cur].p, S.n_db, S.pool.p, S.pool_off.p, S.R, nb, S.dbc[nx].p, S.dbs[nx].p);
        HIPCHK(ctx, hipGetLastError());
        S.cur = nx; S.n_db = tot; S.R += nb;
    }
    return NGSID_OK;
}

extern "C" int32_t ngsid_cluster_greedy(ngsid_ctx* ctx, const ngsid_reads_t* reads, const ngsid_cluster_params_t* prm,
                                        const uint32_t* acc_rank, const int32_t* prev_batch, const double* known_err,
                                        int32_t* rep_of_read, double* hpc_err_out, uint8_t* status_out, uint64_t counters[4])
{
    ApiClock api_clock_(ctx, "cluster_greedy");
    if (!ctx) return NGSID_ERR_ARG;
    if (!reads || !prm || !rep_of_read) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    const int k = prm->k, w = prm->w;
    if (k < 1 || k > NGSID_MAX_K || w < k) NGSID_FAIL(ctx, NGSID_ERR_ARG, "bad k/w (k=%d, w=%d; k <= %d)", k, w, NGSID_MAX_K);
    HostTimer ht(ctx->stream, "cluster");
    DevReads RD; int32_t rc = ngsid_upload_reads(ctx, reads, &RD, true); if (rc) return rc;
    ht.mark("upload");
    const uint64_t N = RD.n;
    if (counters) counters[0] = counters[1] = counters[2] = counters[3] = 0;
    if (N == 0) return NGSID_OK;
    if (N > 0x7fffffffull) NGSID_FAIL(ctx, NGSID_ERR_ARG, "too many reads");
    if (!g_cl_tables[ctx->device & 15]) { HIPCHK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_round2_t), NGSID_ROUND2_T, sizeof(double) * 15)); g_cl_tables[ctx->device & 15] = true; }

    // ---- per-read preprocessing (a1-a3)
    DevBuf<uint64_t>& mzcode = ctx->pol_mzcode; DevBuf<uint32_t>& mzpos = ctx->pol_mzpos;      // compact CSR of the context (12 bytes per minimizer, grow-only), offsets in ctx->mz_off / ctx->h_mzoff
    DevBuf<uint32_t>& mzcnt = ctx->mzc_cnt; DevBuf<uint32_t>& hlen = ctx->mzc_hlen; DevBuf<uint32_t> d_acc; DevBuf<double> herr0, herr, rawerr, d_known; DevBuf<uint8_t> eidx; DevBuf<int> flag;
    ctx->mzc.valid = false;
    HIPCHK(ctx, mzcnt.reserve(N)); HIPCHK(ctx, hlen.reserve(N));
    HIPCHK(ctx, herr0.alloc(N)); HIPCHK(ctx, herr.alloc(N)); HIPCHK(ctx, rawerr.alloc(N)); HIPCHK(ctx, eidx.alloc(N)); HIPCHK(ctx, flag.alloc(2)); HIPCHK(ctx, d_acc.alloc(N));
    HIPCHK(ctx, hipMemsetAsync(flag.p, 0, 2 * sizeof(int), ctx->stream));
    static thread_local PinVec<uint32_t> h_hlen, h_mzcnt;      // host mirrors are kept across calls (fresh multi-megabyte vectors page-fault on every call)
    h_hlen.resize(N); h_mzcnt.resize(N);
    {
        long long bad = -1;
        rc = ngsid_minimizers_csr(ctx, RD, k, w, ngsid_ctx_mz(ctx), mzcnt.p, hlen.p, herr0.p, rawerr.p, h_mzcnt.data(), h_hlen.data(), &bad); if (rc) return rc;
        if (bad >= 0) NGSID_FAIL(ctx, NGSID_ERR_ALPHABET, "read %lld: base outside ACGTN", bad);
    }
    if (k <= 21 && N >= 1024) {       // key of the minimizer cache (the polisher's strand detection may be handed the same reads next); small sets are not worth the fingerprint
        unsigned long long fp = 0; rc = ngsid_reads_fingerprint(ctx, RD, &fp); if (rc) return rc;
        ctx->mzc.n = N; ctx->mzc.total = RD.total; ctx->mzc.k = k; ctx->mzc.w = w; ctx->mzc.fp = fp; ctx->mzc.valid = true;
    }
    if (known_err) { HIPCHK(ctx, d_known.alloc(N)); HIPCHK(ctx, hipMemcpyAsync(d_known.p, known_err, 8 * N, hipMemcpyHostToDevice, ctx->stream)); }
    hipLaunchKernelGGL(k_eidx, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, herr0.p, known_err ? d_known.p : nullptr, N, herr.p, eidx.p);
    HIPCHK(ctx, hipGetLastError());
    if (acc_rank) HIPCHK(ctx, hipMemcpyAsync(d_acc.p, acc_rank, 4 * N, hipMemcpyHostToDevice, ctx->stream));
    else { std::vector<uint32_t> id(N); for (uint64_t i = 0; i < N; ++i) id[i] = (uint32_t)i; HIPCHK(ctx, hipMemcpyAsync(d_acc.p, id.data(), 4 * N, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }

    // ---- max tolerated run of non-shared minimizers per (e1,e2): largest g with p_err^g >= min_prob_no_hits,
    //      p_err^g by left-to-right repeated multiplication exactly like reduce(mul,[p]*g,1)  (cluster.py:97-105)
    std::vector<int32_t> h_maxgap(225);
    for (int i = 0; i < 225; ++i) {
        const double ps = prm->p_shared[i];
        if (ps != ps) { h_maxgap[i] = -2; continue; }
        const double perr = 1.0 - ps; double pr = 1.0; int g = -1;
        for (int t = 0; t < (1 << 20); ++t) { if (pr < prm->min_prob_no_hits) break; g = t; pr = pr * perr; }
        h_maxgap[i] = g;
    }
    DevBuf<int32_t> d_maxgap; HIPCHK(ctx, d_maxgap.alloc(225));
    HIPCHK(ctx, hipMemcpyAsync(d_maxgap.p, h_maxgap.data(), 225 * 4, hipMemcpyHostToDevice, ctx->stream));

    // ---- items (reads to process) and seeded representatives
    int lowest = 1;
    if (prev_batch) { int mn = prev_batch[0]; for (uint64_t i = 1; i < N; ++i) mn = std::min(mn, prev_batch[i]); lowest = std::max(1, mn); }
    static thread_local PinVec<uint32_t> h_items; h_items.clear(); h_items.reserve(N);
    std::vector<uint8_t> h_seeded;
    if (prev_batch) { h_seeded.assign(N, 0); for (uint64_t i = 0; i < N; ++i) { if (prev_batch[i] == lowest) h_seeded[i] = 1; else h_items.push_back((uint32_t)i); } }
    else for (uint64_t i = 0; i < N; ++i) h_items.push_back((uint32_t)i);
    const uint32_t NI = (uint32_t)h_items.size();
    DevBuf<uint32_t> d_items; DevBuf<uint8_t> d_seeded;
    HIPCHK(ctx, d_items.alloc(NI)); if (NI) HIPCHK(ctx, hipMemcpyAsync(d_items.p, h_items.data(), 4ull * NI, hipMemcpyHostToDevice, ctx->stream));
    if (prev_batch) { HIPCHK(ctx, d_seeded.alloc(N)); HIPCHK(ctx, hipMemcpyAsync(d_seeded.p, h_seeded.data(), N, hipMemcpyHostToDevice, ctx->stream)); }

    // ---- state
    DevBuf<int32_t> dec, top, cache_slot, cache_region; DevBuf<uint8_t> kind, alnflag, cache_ptr; DevBuf<uint64_t> cur_hi, cur_lo;
    HIPCHK(ctx, dec.alloc(N)); HIPCHK(ctx, top.alloc(N)); HIPCHK(ctx, cache_slot.alloc(N * NCACHE)); HIPCHK(ctx, cache_region.alloc(N * NCACHE));
    HIPCHK(ctx, kind.alloc(N)); HIPCHK(ctx, alnflag.alloc(N)); HIPCHK(ctx, cache_ptr.alloc(N)); HIPCHK(ctx, cur_hi.alloc(N)); HIPCHK(ctx, cur_lo.alloc(N));
    HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)dec.p, (int)DEC_UNDEC, N, ctx->stream));      // undecided: k_decide_map takes every item with this value
    HIPCHK(ctx, hipMemsetAsync(cache_slot.p, 0xff, 4 * N * NCACHE, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(cache_ptr.p, 0, N, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(alnflag.p, 0, N, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(kind.p, 0, N, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(top.p, 0, 4 * N, ctx->stream));            // (read by rep_can_matter for reads that never reached k_decide_map's top pass: short reads)

    RepStore S; S.ctx = ctx; S.h_pool_off.push_back(0);
    HIPCHK(ctx, S.d_count.alloc(4)); HIPCHK(ctx, S.rep_read.alloc(1024)); HIPCHK(ctx, S.pool_off.alloc(1025)); S.Rcap = 1024;
    HIPCHK(ctx, hipMemsetAsync(S.pool_off.p, 0, 8, ctx->stream));
    HIPCHK(ctx, S.pool.alloc(1 << 16)); HIPCHK(ctx, S.dbc[0].alloc(1 << 16)); HIPCHK(ctx, S.dbs[0].alloc(1 << 16)); HIPCHK(ctx, S.dbc[1].alloc(1 << 16)); HIPCHK(ctx, S.dbs[1].alloc(1 << 16));
    std::vector<uint64_t> h_po;
    if (prev_batch) {       // the seeded representatives, built in chunks (one host round trip per chunk)
        std::vector<uint32_t> seeds;
        for (uint64_t i = 0; i < N; ++i) if (h_seeded[i] && h_hlen[i] >= (uint32_t)k) seeds.push_back((uint32_t)i);
        for (size_t s0 = 0; s0 < seeds.size(); s0 += 512) {
            const uint32_t nch = (uint32_t)std::min<size_t>(512, seeds.size() - s0);
            rc = build_reps(ctx, S, seeds.data() + s0, nch, mzcode.p, ctx->h_mzoff.data(), h_mzcnt.data()); if (rc) return rc;
            h_po.resize(nch + 1);
            HIPCHK(ctx, hipMemcpyAsync(h_po.data(), S.pool_off.p + S.R, 8ull * (nch + 1), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            rc = commit_reps(ctx, S, nch, h_po.data()); if (rc) return rc;
        }
    }

    ClDev D{};
    D.seq = RD.seq; D.qual = RD.qual; D.off = RD.off; D.n = N; D.hlen = hlen.p; D.mzcnt = mzcnt.p; D.herr = herr.p; D.rawerr = rawerr.p; D.eidx = eidx.p; D.accrank = d_acc.p;
    D.mzcode = mzcode.p; D.mzpos = mzpos.p; D.mzoff = ctx->mz_off.p; D.maxgap = d_maxgap.p; D.k = k; D.min_shared = prm->min_shared; D.symmetric = prm->symmetric;
    D.min_fraction = prm->min_fraction; D.mapped_threshold = prm->mapped_threshold; D.aligned_threshold = prm->aligned_threshold;
    D.dec = dec.p; D.kind = kind.p; D.alnflag = alnflag.p; D.top = top.p; D.cur_hi = cur_hi.p; D.cur_lo = cur_lo.p;
    D.cache_slot = cache_slot.p; D.cache_region = cache_region.p; D.cache_ptr = cache_ptr.p; D.errflag = flag.p + 1;

    // speculative block size: adaptive - blocks grow while new representatives are rare (bigger aligner launches, fewer round trips) and
    // shrink when they are frequent (every new representative re-decides the rest of its block)
    // capacity of the per-block buffers = the largest speculative block.  Round 6: 1 M items (262 144 until round 5) - while new representatives are rare a block costs nothing for being
    // large, and every block ends in the tail of a persistent aligner launch (one work item = two pairs = ~2 ms of a wave): fewer, larger launches.  "cluster_block_cap" sets it (tests).
    uint32_t BLK = 1u << 20;
    { const long v = (long)ngsid_opt(ctx, "cluster_block_cap", 0); if (v >= 8192 && v <= (1l << 22)) BLK = (uint32_t)v; }
    uint32_t blk = 32768;
    bool blk_fixed = false;
    { const long v = (long)ngsid_opt(ctx, "cluster_block", 0); if (v >= 64 && v <= (long)BLK) { blk = (uint32_t)v; blk_fixed = true; } }     // dev / test knob: the result must not depend on it
    uint32_t trunc = 32768;                  // cap of a restarting block's rest (below); "cluster_trunc" sets it (0 = off) - tests use small values to cut blocks of small sets
    { const long v = (long)ngsid_opt(ctx, "cluster_trunc", -1); if (v >= 0) trunc = (uint32_t)std::min<long>(v, 1l << 22); }
    DevBuf<uint64_t>& cnt = ctx->cl_cnt; uint32_t stride = 0;      // (kept in the context: a block's matrix is up to a few GB, a fresh allocation of it costs 20+ ms)
    DevBuf<uint32_t> req_q, req_t, req_slot, d_scal; DevBuf<int32_t> req_open, req_mid, req_region;
    HIPCHK(ctx, req_q.alloc(BLK)); HIPCHK(ctx, req_t.alloc(BLK)); HIPCHK(ctx, req_slot.alloc(BLK)); HIPCHK(ctx, req_open.alloc(BLK)); HIPCHK(ctx, req_mid.alloc(BLK)); HIPCHK(ctx, req_region.alloc(BLK));
    HIPCHK(ctx, d_scal.alloc(4)); HIPCHK(ctx, hipMemsetAsync(d_scal.p, 0, 16, ctx->stream));
    uint32_t tcur = 1;
    constexpr uint32_t TMAX = 64;             // new representatives committed per pass (one lane per tentative column in k_first_affected)
    DevBuf<unsigned long long> d_mask; HIPCHK(ctx, d_mask.alloc(BLK / 64 + 1)); PinVec<unsigned long long> h_mask(BLK / 64 + 1);
    auto refresh = [&]() { D.rep_read = S.rep_read.p; D.pool = S.pool.p; D.pool_off = S.pool_off.p; };

    ht.mark("setup");
    for (uint32_t b0 = 0; b0 < NI; ) {
        uint32_t b1 = std::min<uint32_t>(NI, b0 + blk);
        uint32_t rounds = 0;                      // restart rounds of this block
        uint32_t lo = b0;                         // first uncommitted item
        uint32_t newreps = 0;
        uint64_t aln_first = 0, aln_total = 0; bool first_round = true;      // pairs aligned in the block's first alignment round / in all of them (round 6: the block size follows the share of RE-alignments)
        bool need_full = true;
        while (lo < b1) {
            refresh();
            if (need_full) {
                // hit matrix for items [lo,b1) against the whole index; stride leaves room for new representatives
                const uint32_t want = ((S.R + 256 + 63) / 64) * 64;
                // rows = the items of THIS block (blocks shrink when representatives are frequent, so rows x columns stays moderate: a noisy read
                // set with 50 k representatives would otherwise ask for BLK x 50 k x 8 B = 100 GB)
                stride = std::max(stride, want);
                { const uint64_t need = (uint64_t)(b1 - b0) * stride; if (!cnt.p || cnt.cap < need) HIPCHK(ctx, cnt.alloc(need + need / 4)); }
                HIPCHK(ctx, hipMemsetAsync(cnt.p, 0, 8ull * (uint64_t)(b1 - b0) * stride, ctx->stream));
                { ProfScope ps_(ctx, "k_count_hits"); hipLaunchKernelGGL(k_count_hits, dim3((b1 - lo + 3) / 4), dim3(256), 0, ctx->stream, D, d_items.p, lo, b1, b0,
                                   S.dbc[S.cur].p, S.dbs[S.cur].p, 0u, S.n_db, cnt.p, stride); }
                HIPCHK(ctx, hipGetLastError());
                need_full = false;
            }
            { ProfScope ps_(ctx, "k_decide_map"); hipLaunchKernelGGL(k_decide_map, dim3((b1 - lo + 3) / 4), dim3(256), 0, ctx->stream, D, d_items.p, lo, b1, b0, cnt.p, stride, S.R); }
            HIPCHK(ctx, hipGetLastError());
            const uint32_t w0 = (lo - b0) >> 6, w1 = (b1 - b0 + 63) >> 6;
            int eflag = 0;
            for (;;) {      // alignment rounds: resolve from cache or request pairs, align, cache, repeat  (d_scal[0] is zero here: set at the start, reset by k_newrep_mask)
                { ProfScope ps_(ctx, "k_aln_next"); hipLaunchKernelGGL(k_aln_next, dim3((b1 - lo + 15) / 16), dim3(1024), 0, ctx->stream, D, d_items.p, lo, b1, b0, cnt.p, stride, S.R,
                                   req_q.p, req_t.p, req_slot.p, req_open.p, req_mid.p, d_scal.p); }
                HIPCHK(ctx, hipGetLastError());
                // the new-representative mask of the block is computed and fetched in the SAME round trip as the request count: when no pair is requested every item
                // of [lo, b1) is decided and the mask is the one the next step needs (most rounds of a noisy tail; one host round trip less per round); when pairs are
                // requested it is recomputed after the alignments
                uint32_t nreq = 0;
                hipLaunchKernelGGL(k_newrep_mask, dim3(((w1 - w0) * 64 + 255) / 256), dim3(256), 0, ctx->stream, dec.p, d_items.p, b0 + w0 * 64, lo, b1, b0, d_mask.p, w1, d_scal.p, flag.p + 1);
                HIPCHK(ctx, hipGetLastError());
                HIPCHK(ctx, hipMemcpyAsync(h_mask.data() + w0, d_mask.p + w0, 8ull * (w1 - w0 + 1), hipMemcpyDeviceToHost, ctx->stream));      // mask words + the tail word (nreq | eflag << 32)
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                nreq = (uint32_t)(h_mask[w1] & 0xffffffffull); eflag = (int)(h_mask[w1] >> 32);
                if (!nreq) break;
                aln_total += nreq; if (first_round) aln_first += nreq;
                AlignJob J{};
                J.qseq = RD.seq; J.qoff = RD.off; J.tseq = RD.seq; J.toff = RD.off; J.qidx = req_q.p; J.tidx = req_t.p; J.npairs = nreq;
                J.match = 2; J.mismatch = -2; J.ext = 1; J.k = k; J.open = req_open.p; J.match_id = req_mid.p;     // cluster.py:130
                J.score = nullptr; J.ncols = nullptr; J.nmatch = nullptr; J.region = req_region.p; J.bp = nullptr; J.bp_windows = 0; J.window = 1; J.span = nullptr;
                rc = ngsid_launch_align(ctx, J, RD.maxlen, RD.maxlen, 5, RD.minlen); if (rc) return rc;     // gap open is 2..5 (cluster.py:189-196)
                hipLaunchKernelGGL(k_cache_insert, dim3((nreq + 255) / 256), dim3(256), 0, ctx->stream, D, req_q.p, req_slot.p, req_region.p, nreq);
                HIPCHK(ctx, hipGetLastError());
            }
            first_round = false;
            // ---- the items that decided "new representative", in block order (bit mask of the block, fetched with the last request count above, scanned on the host)
            if (eflag) NGSID_FAIL(ctx, NGSID_ERR_NO_PTABLE, "no p_shared entry for an (e1,e2) pair met during mapping (KeyError in cluster.py:367)");
            uint32_t C[TMAX + 1]; uint32_t nC = 0;
            for (uint32_t wd = w0; wd < w1 && nC <= TMAX; ++wd) {
                unsigned long long m = h_mask[wd];
                while (m && nC <= TMAX) { C[nC++] = b0 + wd * 64 + (uint32_t)__builtin_ctzll(m); m &= m - 1; }
            }
            if (nC == 0) { lo = b1; break; }
            // ---- up to TMAX of them become representatives at once.  All are built (slots R, R+1, ...) and the later items are counted against
            // each; everything before the first item that shares >= min_shared minimizers with an EARLIER tentative representative is final
            // (such an item is the only kind whose decision can change), the tentative representatives before that item are committed, the
            // rest of the block is decided again.  With one new representative per pass this is the plain greedy restart.
            const uint32_t T = std::min<uint32_t>(std::min<uint32_t>(tcur, nC), stride - S.R);
            const uint32_t next_c = nC > T ? C[T] : 0xffffffffu;
            uint32_t creads[TMAX];
            for (uint32_t t = 0; t < T; ++t) creads[t] = h_items[C[t]];
            rc = build_reps(ctx, S, creads, T, mzcode.p, ctx->h_mzoff.data(), h_mzcnt.data()); if (rc) return rc;
            refresh();
            if (C[0] + 1 < b1) {
                PosBatch PB; for (uint32_t t = 0; t < 64; ++t) PB.pos[t] = t < T ? C[t] : 0xffffffffu;
                ProfScope ps_(ctx, "k_count_hits");
                hipLaunchKernelGGL(k_count_hits_reps, dim3((b1 - C[0] - 1 + 3) / 4, T), dim3(256), 0, ctx->stream, D, d_items.p, C[0] + 1, b1, b0, S.R, PB, cnt.p, stride);
            }
            HIPCHK(ctx, hipGetLastError());
            uint32_t cut = 0xffffffffu;
            HIPCHK(ctx, hipMemsetAsync(d_scal.p + 1, 0xff, 4, ctx->stream));
            if (C[0] + 1 < b1) hipLaunchKernelGGL(k_first_affected, dim3((b1 - C[0] - 1 + 3) / 4), dim3(256), 0, ctx->stream, D, d_items.p, C[0] + 1, b1, b0, cnt.p, stride, S.R, T, d_scal.p + 1);
            HIPCHK(ctx, hipGetLastError());
            h_po.resize(T + 1);
            HIPCHK(ctx, hipMemcpyAsync(&cut, d_scal.p + 1, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(h_po.data(), S.pool_off.p + S.R, 8ull * (T + 1), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            cut = std::min(cut, next_c);
            uint32_t c = 0; while (c < T && C[c] < cut) ++c;                 // >= 1: the affected item lies behind C[0]
            const uint32_t R_before = S.R;
            rc = commit_reps(ctx, S, c, h_po.data()); if (rc) return rc;
            newreps += c;
            tcur = c == T ? std::min<uint32_t>(TMAX, tcur * 2) : std::max<uint32_t>(1, c);      // speculate wider only while it pays
            if (cut >= b1) { lo = b1; break; }
            lo = cut;
            // the columns of the tentative representatives that were not committed go back to zero (their entries sit in rows >= cut)
            if (c < T) HIPCHK(ctx, hipMemset2DAsync(cnt.p + (uint64_t)(lo - b0) * stride + S.R, (size_t)stride * 8, 0, (size_t)(T - c) * 8, b1 - lo, ctx->stream));
            refresh();
            if (S.R + 1 > stride) need_full = true;
            hipLaunchKernelGGL(k_reset_items, dim3((b1 - lo + 255) / 256), dim3(256), 0, ctx->stream, D, d_items.p, lo, b1, b0, cnt.p, stride, R_before, c);
            HIPCHK(ctx, hipGetLastError());
            // Round 6: a block that keeps restarting is cut short.  Every restart round runs five kernels over the rest of the block (reset, decide, alignment cursor, mask, hit
            // counts of the next tentative representatives), so its cost grows with that rest, and the NUMBER of rounds is given by the data (one per stretch between dependent
            // new representatives) - the noisy tail of a score-ordered set brings thousands of them at once (mu = 14: 2 196 new representatives in the last 408 k reads, after
            // 540 k reads with none, i.e. in a block that had grown to its largest size).  From the third round on the rest is capped at `trunc` items; the items behind the cut
            // go back to "undecided" and are decided from scratch by the next block, which starts at that size.  Results do not depend on where blocks end (tested).
            ++rounds;
            if (trunc && rounds >= 3 && b1 - lo > trunc) {
                const uint32_t nb1 = lo + trunc;
                hipLaunchKernelGGL(k_undecide, dim3((b1 - nb1 + 255) / 256), dim3(256), 0, ctx->stream, D, d_items.p, nb1, b1);
                HIPCHK(ctx, hipGetLastError());
                b1 = nb1; if (!blk_fixed) blk = std::max<uint32_t>(trunc, 8192);
            }
        }
        b0 = b1;
        if (!blk_fixed) {
            // blocks grow while new representatives are rare and shrink when they are frequent (every new representative re-decides part of the rest of its block)
            const double waste = aln_first ? (double)(aln_total - aln_first) / (double)aln_first : (aln_total ? 1.0 : 0.0);
            static const bool trace_blocks = getenv("NGSID_CLUSTER_TRACE") != nullptr;      // dev aid: one line per block on stderr
            if (trace_blocks) fprintf(stderr, "[ngsid cluster] block ending at %u: %u restart rounds, %u new representatives (R = %u), pairs aligned in the first round %llu, later %llu (%.3f)\n", b1, rounds, newreps, S.R, (unsigned long long)aln_first, (unsigned long long)(aln_total - aln_first), waste);
            // (a rule by restart ROUNDS instead of new representatives was measured in round 6 and lost: 0.45 against 0.41 s at mu = 14 - small blocks stay the better ones while
            // representatives keep coming, whatever the number of rounds they take)
            const long s_reps = (long)ngsid_opt(ctx, "cluster_shrink_reps", 32); const uint32_t floor_blk = (uint32_t)ngsid_opt(ctx, "cluster_block_floor", 4096);
            if (newreps <= 2) blk = std::min<uint32_t>(blk * 2, BLK); else if ((long)newreps > s_reps) blk = std::max<uint32_t>(blk / 2, floor_blk);
            // the hit matrix of a block is rows x (representatives + room) x 8 bytes: a noisy set with thousands of representatives keeps its blocks under 12 GB of it
            const uint64_t cols = std::max<uint64_t>(stride, ((uint64_t)S.R + 256 + 63) / 64 * 64);
            while (blk > 8192 && (uint64_t)blk * cols * 8 > ((uint64_t)12 << 30)) blk /= 2;
        }
    }
    ht.mark("blocks");
    // ---- results
    DevBuf<int32_t> d_rep; DevBuf<uint8_t> d_status; DevBuf<double> d_herr_out; DevBuf<unsigned long long> d_counters;
    HIPCHK(ctx, d_rep.alloc(N)); HIPCHK(ctx, d_status.alloc(N)); HIPCHK(ctx, d_herr_out.alloc(N)); HIPCHK(ctx, d_counters.alloc(4));
    HIPCHK(ctx, hipMemsetAsync(d_counters.p, 0, 32, ctx->stream));
    refresh();
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, D, N, prev_batch ? d_seeded.p : nullptr, d_rep.p, d_status.p, d_herr_out.p, d_counters.p);
    HIPCHK(ctx, hipGetLastError());
    static thread_local PinVec<uint8_t> h_status; static thread_local PinVec<double> h_herr; static thread_local PinVec<int32_t> h_rep; h_status.resize(N); h_herr.resize(N); h_rep.resize(N); unsigned long long h_cnt[4];
    HIPCHK(ctx, hipMemcpyAsync(h_rep.data(), d_rep.p, 4 * N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_status.data(), d_status.p, N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_herr.data(), d_herr_out.p, 8 * N, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_cnt, d_counters.p, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ht.mark("finalize + download");
    memcpy(rep_of_read, h_rep.data(), 4 * N);
    if (status_out) memcpy(status_out, h_status.data(), N);
    if (hpc_err_out) memcpy(hpc_err_out, h_herr.data(), 8 * N);
    if (counters) for (int i = 0; i < 4; ++i) counters[i] = h_cnt[i];
    return NGSID_OK;
}
