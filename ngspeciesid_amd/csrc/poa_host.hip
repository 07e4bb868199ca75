// poa_host.hip - host drivers over the POA tile engine: the depth-tiled hierarchy, ngsid_poa_consensus (a13,a14)
// and ngsid_polish (a16,a17).  Mirrors oracle/ngsid_oracle_poa.c: run_hierarchy / ongsid_poa_consensus / ongsid_polish.
#include "ngsid_internal.h"
#include "k_poa.h"
#include <thread>
#include <atomic>
#include <chrono>
#include <algorithm>
#include <vector>
#include <string>
#include <memory>

namespace {
#define HT_INIT

struct Unit {                       // one consensus problem: a cluster (spoa stage) or a backbone window (polish)
    std::vector<uint32_t> seqs;     // indices into the CURRENT level's PSeq array, in order
    int bb = -1;                    // backbone index (into the bbs array) or -1
    bool done = false;
    uint32_t single_maxlen = 0;     // round 6 (single_below): set (> 0) by the caller for a unit that runs as ONE graph = the longest sequence that can enter it (reads of its members / layers, its backbone)
    std::string result; std::vector<uint32_t> cov; bool has_result = false;
};

struct Level {                      // device buffers of one hierarchy level (kept alive while the next level reads them)
    DevBuf<PSeq> seqs; DevBuf<uint8_t> out; DevBuf<int32_t> out_len; DevBuf<uint64_t> out_cw; DevBuf<uint32_t> out_n, out_cov;
};

__global__ void k_make_pseq_reads(const uint8_t* seq, const uint8_t* qual, const uint64_t* off, uint64_t n, int mode, const uint32_t* weight, PSeq* out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PSeq S; S.s = seq + off[i]; S.q = qual ? qual + off[i] : nullptr; S.len = (int32_t)(off[i + 1] - off[i]); S.uw = 1; S.cw = 1; S.mode = mode; S.a0 = 0; S.a1 = -1;
    if (weight) {        // a sequence that stands for weight[i] reads (ngsid_poa_consensus_weighted): unit weights of that size, like the tile consensuses of the upper levels
        const uint32_t wv = weight[i]; S.q = nullptr; S.cw = wv; S.uw = (int32_t)(wv > (1u << 20) ? (1u << 20) : (wv < 1u ? 1u : wv));
    }
    out[i] = S;
}

struct HierParams { int m, n, g, band, node_cap, D, upper_mode; bool want_cov; int trim_tiles; int single_below = 0; };      // single_below: ngsid_poa_params_t.single_below (units the caller marked with single_maxlen)

// Runs all units to completion.  level0: device PSeq array (nseq0 entries) whose max length is maxlen0;
// bbs: device backbone PSeqs (may be null), maxbb = longest backbone.
int32_t run_hierarchy_host(ngsid_ctx* ctx, const PSeq* d_level0, uint32_t maxlen0, const PSeq* d_bbs, const std::vector<int>& bb_len,
                           std::vector<Unit>& units, const HierParams& hp)
{
    const PSeq* cur = d_level0; uint32_t cur_maxlen = maxlen0;
    int slots_cap = 0;
    HostTimer ht(ctx->stream, "hierarchy");
    for (int level = 0;; ++level) {
        // ---- jobs of this level
        // host lists of a level: kept across levels and calls (a million entries per level; fresh allocations would page-fault every time)
        static thread_local PinVec<uint32_t> job_off, seq_idx; static thread_local std::vector<uint32_t> job_unit; static thread_local PinVec<int32_t> job_bb;
        job_off.clear(); job_off.push_back(0); seq_idx.clear(); job_unit.clear(); job_bb.clear();
        static thread_local PinVec<uint8_t> job_final; job_final.clear();      // trim 3 (hp.trim_tiles & 4): tiles that end their unit
        { size_t tot = 0; for (const Unit& U : units) if (!U.done) tot += U.seqs.size(); seq_idx.reserve(tot); }
        uint32_t maxD = 0; int maxL0 = 1; bool any_nobb = false;
        for (size_t u = 0; u < units.size(); ++u) {
            Unit& U = units[u]; if (U.done) continue;
            const uint32_t ncur = (uint32_t)U.seqs.size();
            if (ncur == 0) { U.done = true; continue; }
            const uint32_t Dl = hp.D > 0 ? (uint32_t)hp.D : ncur; const uint32_t nt = poa_ntiles(ncur, Dl);
            for (uint32_t t = 0; t < nt; ++t) {
                const uint32_t a = t * Dl, b = t + 1 == nt ? ncur : a + Dl;
                for (uint32_t x = a; x < b; ++x) seq_idx.push_back(U.seqs[x]);
                job_off.push_back((uint32_t)seq_idx.size()); job_bb.push_back(U.bb); job_unit.push_back((uint32_t)u); job_final.push_back(nt == 1 ? 1 : 0);
                maxD = std::max(maxD, b - a);
            }
            if (U.bb >= 0) maxL0 = std::max(maxL0, bb_len[U.bb]); else any_nobb = true;
        }
        const uint32_t njobs = (uint32_t)job_unit.size();
        if (njobs == 0) break;
        if (level == 0) ht.mark("L0 host job lists");
        if (any_nobb) maxL0 = std::max<int>(maxL0, (int)cur_maxlen);   // without a backbone the first member sets L0 (<= the longest member)
        // capacity (LDS sizing): the largest per-job capacity the oracle rule can produce
        long long capV = (long long)maxL0 * (hp.node_cap > 0 ? hp.node_cap : 28) / 16; capV = std::max<long long>(capV, maxL0 + 64); capV = std::max<long long>(capV, (long long)cur_maxlen + 1);
        capV = (capV + 7) & ~7ll;
        const int Lmax = (int)((std::max<uint32_t>(cur_maxlen, 1) + 15) & ~15u);
        ngsid_ctx::PoaLevelBufs* Lv = &ctx->poa_lv[level & 1];      // level L+1 reads what level L wrote: two alternating, grow-only buffer sets
        DevBuf<uint32_t>& d_job_off = Lv->job_off; DevBuf<uint32_t>& d_seq_idx = Lv->seq_idx; DevBuf<uint32_t>& d_flags = Lv->flags; DevBuf<int32_t>& d_job_bb = Lv->job_bb;
        HIPCHK(ctx, d_job_off.reserve(job_off.size())); HIPCHK(ctx, d_seq_idx.reserve(seq_idx.size())); HIPCHK(ctx, d_job_bb.reserve(job_bb.size())); HIPCHK(ctx, d_flags.reserve(4));
        HIPCHK(ctx, hipMemcpyAsync(d_job_off.p, job_off.data(), 4 * job_off.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(d_seq_idx.p, seq_idx.data(), 4 * seq_idx.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(d_job_bb.p, job_bb.data(), 4 * job_bb.size(), hipMemcpyHostToDevice, ctx->stream));
        if (hp.trim_tiles & 4) { HIPCHK(ctx, Lv->job_final.reserve(job_final.size())); HIPCHK(ctx, hipMemcpyAsync(Lv->job_final.p, job_final.data(), job_final.size(), hipMemcpyHostToDevice, ctx->stream)); }
        int slots = slots_cap ? slots_cap : (int)std::min<uint32_t>(maxD, (uint32_t)std::max<long long>(1, ngsid_opt(ctx, "poa_out_slots", 4)));
        static thread_local PinVec<uint32_t> h_out_n; static thread_local PinVec<int32_t> h_out_len, h_out_span; static thread_local PinVec<uint64_t> h_out_cw;
        const bool need_cov = hp.want_cov || hp.trim_tiles;
        for (;;) {      // retry with more output slots if a tile had to split more often than `slots`
            HIPCHK(ctx, Lv->out.reserve((size_t)njobs * slots * capV)); HIPCHK(ctx, Lv->out_len.reserve((size_t)njobs * slots)); HIPCHK(ctx, Lv->out_cw.reserve((size_t)njobs * slots)); HIPCHK(ctx, Lv->out_n.reserve(njobs)); HIPCHK(ctx, Lv->out_span.reserve((size_t)njobs * slots * 2));
            if (need_cov) HIPCHK(ctx, Lv->out_cov.reserve((size_t)njobs * slots * capV));
            HIPCHK(ctx, hipMemsetAsync(d_flags.p, 0, 16, ctx->stream));
            PoaJobSet J{};
            J.seqs = cur; J.bbs = d_bbs; J.seq_idx = d_seq_idx.p; J.job_off = d_job_off.p; J.job_bb = d_job_bb.p; J.njobs = njobs;
            J.m = hp.m; J.n = hp.n; J.g = hp.g; J.Vcap = (int)capV; J.Ecap = (int)(3 * capV / 2); J.Lmax = Lmax; J.D = slots; J.node_cap = hp.node_cap; J.trim_tiles = (hp.trim_tiles & 1) | ((level > 0 && (hp.trim_tiles & 1)) ? 2 : 0); J.job_final = (hp.trim_tiles & 4) ? Lv->job_final.p : nullptr;
            J.out = Lv->out.p; J.out_len = Lv->out_len.p; J.out_cw = Lv->out_cw.p; J.out_n = Lv->out_n.p; J.out_cov = need_cov ? Lv->out_cov.p : nullptr; J.out_span = Lv->out_span.p;
            J.dropped = d_flags.p; J.slot_overflow = d_flags.p + 1;
            DevBuf<unsigned long long> d_ph; static const bool want_ph = getenv("NGSID_POA_PHASES") != nullptr;
            if (want_ph) { HIPCHK(ctx, d_ph.alloc(24 * 256)); HIPCHK(ctx, hipMemsetAsync(d_ph.p, 0, 192 * 256, ctx->stream)); J.phase_cycles = d_ph.p; J.phase_detail = atoi(getenv("NGSID_POA_PHASES")) >= 2; }
            if (level == 0) ht.mark("L0 alloc + upload");
            int32_t rc = poa_run_jobs(ctx, J, hp.band); if (rc) return rc;
            if (level == 0) ht.mark("L0 kernel");
            uint32_t h_flags[4];
            h_out_n.resize(njobs); h_out_len.resize((size_t)njobs * slots); h_out_cw.resize((size_t)njobs * slots); h_out_span.resize((size_t)njobs * slots * 2);
            auto download = [&]() -> int32_t {
                HIPCHK(ctx, hipMemcpyAsync(h_flags, d_flags.p, 16, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(h_out_n.data(), Lv->out_n.p, 4ull * njobs, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(h_out_len.data(), Lv->out_len.p, 4ull * njobs * slots, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(h_out_cw.data(), Lv->out_cw.p, 8ull * njobs * slots, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipMemcpyAsync(h_out_span.data(), Lv->out_span.p, 8ull * njobs * slots, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                return NGSID_OK;
            };
            rc = download(); if (rc) return rc;
            // band-edge check (oracle run_tile): tiles in which a traceback touched a clipped edge of its band run again, whole, with twice the band
            for (int bw = hp.band <= 64 ? 64 : (hp.band <= 128 ? 128 : 256); bw < 256 && !h_flags[1] && !h_flags[2];) {
                static thread_local PinVec<uint32_t> redo; redo.clear();
                for (uint32_t j = 0; j < njobs; ++j) if (h_out_n[j] & 0x80000000u) redo.push_back(j);
                if (redo.empty()) break;
                bw *= 2;
                HIPCHK(ctx, Lv->job_list.reserve(redo.size()));
                HIPCHK(ctx, hipMemcpyAsync(Lv->job_list.p, redo.data(), 4 * redo.size(), hipMemcpyHostToDevice, ctx->stream));
                J.job_list = Lv->job_list.p; J.nrun = (uint32_t)redo.size();
                rc = poa_run_jobs(ctx, J, bw); if (rc) return rc;
                rc = download(); if (rc) return rc;
                ctx->poa_redo_tiles += redo.size();
            }
            for (uint32_t j = 0; j < njobs; ++j) h_out_n[j] &= 0x7fffffffu;
            if (want_ph) { std::vector<unsigned long long> hs(24 * 256); HIPCHK(ctx, hipMemcpy(hs.data(), d_ph.p, 192 * 256, hipMemcpyDeviceToHost)); unsigned long long h[24] = {0}; for (int s_ = 0; s_ < 256; ++s_) for (int k_ = 0; k_ < 24; ++k_) h[k_] += hs[s_ * 24 + k_]; fprintf(stderr, "[ngsid poa phases, Mcycles] jobs %u prepass %.1f forward %.1f traceback %.1f update %.1f emit %.1f | rows %llu non-chain %llu | sums: bestv %llu bestpk %llu nnew %llu alnsum %llu outlen %llu | tb iters %llu reloads %llu reload Mcycles %.1f | emit backtrack %.1f | row kinds: tight %llu in %llu runs, chain %llu, near %llu, generic %llu | update: A %.1f S+D %.1f N %.1f\n", njobs, h[0] / 1e6, h[1] / 1e6, h[2] / 1e6, h[3] / 1e6, h[4] / 1e6, h[5], h[6], h[8], h[9], h[10], h[11], h[12], h[7], h[13], h[14] / 1e6, h[15] / 1e6, h[16], h[17], h[18], h[19], h[20], h[21] / 1e6, h[22] / 1e6, h[23] / 1e6); }
            if (h_flags[2]) NGSID_FAIL(ctx, NGSID_ERR_HIP, "internal: POA tile kernel loop guard tripped (code %u)", h_flags[2]);
            if (!h_flags[1]) break;
            if (slots >= (int)maxD) NGSID_FAIL(ctx, NGSID_ERR_HIP, "internal: POA output slot overflow at full depth");
            slots = (int)std::min<uint32_t>(maxD, (uint32_t)slots * 4); slots_cap = slots;
        }
        if (level == 0) ht.mark("L0 download");
        // ---- distribute outputs to units
        std::vector<std::vector<uint32_t>> outs(units.size());         // flat slot indices (job*slots + s) in job order
        for (uint32_t j = 0; j < njobs; ++j) for (uint32_t s = 0; s < h_out_n[j]; ++s) outs[job_unit[j]].push_back(j * (uint32_t)slots + s);
        static thread_local PinVec<PSeq> next; uint32_t next_maxlen = 0;
        { size_t tot = 0; for (uint32_t j = 0; j < njobs; ++j) tot += h_out_n[j]; next.clear(); next.reserve(tot); }
        for (size_t u = 0; u < units.size(); ++u) {
            Unit& U = units[u]; if (U.done) continue;
            const std::vector<uint32_t>& O = outs[u];
            const size_t ncur = U.seqs.size();
            int pick = -1;
            if (O.empty()) { U.done = true; continue; }
            if (O.size() == 1) pick = 0;
            else if (O.size() >= ncur) { pick = 0; for (size_t i = 1; i < O.size(); ++i) if (h_out_cw[O[i]] > h_out_cw[O[pick]]) pick = (int)i; }
            if (pick >= 0) {
                const uint32_t sl = O[pick]; const int len = h_out_len[sl];
                U.result.resize(len);
                if (len) HIPCHK(ctx, hipMemcpy(&U.result[0], Lv->out.p + (size_t)sl * capV, len, hipMemcpyDeviceToHost));
                if (hp.want_cov) { U.cov.resize(len); if (len) HIPCHK(ctx, hipMemcpy(U.cov.data(), Lv->out_cov.p + (size_t)sl * capV, 4ull * len, hipMemcpyDeviceToHost)); }
                U.has_result = true; U.done = true; continue;
            }
            U.seqs.clear();
            for (uint32_t sl : O) {
                const uint64_t cw = h_out_cw[sl];
                PSeq S; S.s = Lv->out.p + (size_t)sl * capV; S.q = nullptr; S.len = h_out_len[sl];
                S.uw = cw > (1u << 20) ? (1 << 20) : (int)cw; if (S.uw < 1) S.uw = 1;
                S.cw = (uint32_t)(cw > 0xffffffffull ? 0xffffffffull : cw); S.mode = hp.upper_mode; S.a0 = 0; S.a1 = -1;
                if (U.bb >= 0) {     // a tile consensus is a layer of the window like the reads it stands for (oracle run_hierarchy): global only if it spans the window
                    const int wlen = bb_len[U.bb], offset = (int)(0.01 * (double)wlen), begin = h_out_span[2 * (size_t)sl], end = h_out_span[2 * (size_t)sl + 1];
                    if (end >= begin) { S.a0 = begin; S.a1 = end; S.mode = (begin < offset && end > wlen - offset) ? NGSID_POA_GLOBAL : NGSID_POA_SEMI; }
                }
                U.seqs.push_back((uint32_t)next.size()); next.push_back(S); next_maxlen = std::max<uint32_t>(next_maxlen, (uint32_t)S.len);
            }
        }
        if (level == 0) ht.mark("L0 distribute"); else { char lb[48]; snprintf(lb, sizeof lb, "level %d (%u tiles)", level, njobs); ht.mark(lb); }
        if (next.empty()) break;
        HIPCHK(ctx, Lv->seqs.reserve(sizeof(PSeq) * next.size()));
        HIPCHK(ctx, hipMemcpyAsync(Lv->seqs.p, next.data(), sizeof(PSeq) * next.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        cur = (const PSeq*)Lv->seqs.p; cur_maxlen = next_maxlen;
    }
    ht.mark("levels >= 1");
    return NGSID_OK;
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Device-driven hierarchy (round 3).  The host-driven loop above synchronises after every level (flags, output counts, lengths, weights and
// spans come back, the host builds the next level's sequence / tile lists and uploads them): 4 - 5 ms of idle GPU per polishing iteration at
// 1 M reads, and the top levels are a string of millisecond launches with a round trip in between.  Here the level bookkeeping runs in small
// kernels on the same stream - band-edge redo lists, per-unit output scan, next-level sequence descriptors, next-level tile lists, results
// of finished units copied to an arena - and the tile kernels read their tile count from device memory.  The host enqueues all levels of a
// hierarchy at once and synchronises ONCE.  Same results as the host-driven loop (tests run both: ngsid_ctx_option "poa_host_levels").
// Anything the fixed launch geometry cannot hold (a tile with more outputs than `slots`, a tile consensus much longer than the longest input,
// more upper-level tiles than planned) sets a flag and the call is repeated by the host-driven loop.
enum { C_NJOBS = 0 /* 2 words: level parity */, C_NSEQ = 2, C_OVERFLOW = 3, C_RES_USED = 4, C_REDO_TOTAL = 5, C_FLAGS = 8 /* dropped, slot_overflow (2 words) */, C_REDO = 16 /* + 2 * level + stage */, C_MAXLV = 48, C_WORK = C_REDO + 2 * C_MAXLV /* + 3 * level + instance */, C_WORDS = C_WORK + 3 * C_MAXLV };

struct HierDev {
    uint32_t U; int D, slots, capV, upper_mode, want_cov, Lmax, keep_final; uint32_t cap_jobs[2];
    const int32_t* unit_bb; const int32_t* unit_wlen;
    uint32_t* unit_ncur; uint32_t* unit_job0[2]; uint32_t* unit_njobs[2]; uint32_t* unit_seq0; int32_t* unit_pick; uint32_t* unit_tmp /* njobs_next, nseq_next, res_len of this level: 3 U */;
    uint32_t* res_off; int32_t* res_len; uint8_t* res; uint32_t* res_cov;
    uint32_t* ctrl;
};
struct LevelDev { PSeq* seqs; uint8_t* out; int32_t* out_len; int32_t* out_span; uint64_t* out_cw; uint32_t* out_n; uint32_t* out_cov; uint32_t* job_off; int32_t* job_bb; uint32_t* job_unit; uint32_t* job_pos; uint32_t* job_list; uint8_t* job_final; };

// tiles whose traceback touched a clipped band edge (bit 31 of out_n) -> list for the launch with twice the band
__global__ __launch_bounds__(256) void k_poa_redo_list(const uint32_t* __restrict__ out_n, const uint32_t* __restrict__ njobs, uint32_t* __restrict__ list, uint32_t* __restrict__ cnt, uint32_t* __restrict__ total)
{
    const uint32_t n = *njobs; const int lane = threadIdx.x & 63;
    for (uint32_t j0 = blockIdx.x * blockDim.x; j0 < n; j0 += gridDim.x * blockDim.x) {
        const uint32_t j = j0 + threadIdx.x;
        const bool f = j < n && (out_n[j] & 0x80000000u);
        const unsigned long long m = __ballot(f);
        if (!m) continue;
        const int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0; if (lane == leader) { base = atomicAdd(cnt, (uint32_t)__popcll(m)); atomicAdd(total, (uint32_t)__popcll(m)); }
        base = __shfl(base, leader);
        if (f) list[base + __popcll(m & ((1ull << lane) - 1))] = j;
    }
}

__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* sm /* 8 words */, uint32_t& total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t y = __shfl_up(x, d); if (lane >= d) x += y; }
    __syncthreads();
    if (lane == 63) sm[w] = x;
    __syncthreads();
    uint32_t off = 0; for (int k = 0; k < w; ++k) off += sm[k];
    total = sm[0] + sm[1] + sm[2] + sm[3];
    return off + x - v;
}

// one workgroup per unit: output count of the level, position of every tile's outputs among them, fate of the unit
//   pick: -3 = was finished before, -2 = goes on, -1 = finishes without a result, >= 0 = finishes with the output in that slot
__global__ __launch_bounds__(256) void k_poa_unit_scan(HierDev H, LevelDev Lv, int par)
{
    __shared__ uint32_t sm[8]; __shared__ unsigned long long best_s[4]; __shared__ uint32_t one_slot;
    const uint32_t u = blockIdx.x; if (u >= H.U || H.ctrl[C_OVERFLOW]) return;
    const uint32_t j0 = H.unit_job0[par][u], nj = H.unit_njobs[par][u];
    uint32_t* tmp = H.unit_tmp + 3ull * u;
    if (nj == 0) {        // finished at an earlier level, or empty from the start
        if (threadIdx.x == 0) { if (H.unit_pick[u] == -2) H.unit_pick[u] = -1; else if (H.unit_pick[u] >= 0 || H.unit_pick[u] == -1) H.unit_pick[u] = -3; tmp[0] = 0; tmp[1] = 0; tmp[2] = 0; }
        return;
    }
    if (threadIdx.x == 0) one_slot = 0xffffffffu;
    __syncthreads();
    uint32_t running = 0;
    for (uint32_t b = 0; b < nj; b += 256) {
        const uint32_t x = b + threadIdx.x; const uint32_t j = j0 + x;
        const uint32_t c = x < nj ? (Lv.out_n[j] & 0x7fffffffu) : 0u;
        uint32_t tot; const uint32_t ex = block_excl_scan_256(c, sm, tot);
        if (x < nj) Lv.job_pos[j] = running + ex;
        if (c) one_slot = j * (uint32_t)H.slots;                  // only read when the unit has exactly one output
        running += tot;
        __syncthreads();
    }
    const uint32_t T = running, ncur = H.unit_ncur[u];
    int pick = -2;
    if (T == 0) pick = -1;
    else if (T == 1) { __syncthreads(); pick = (int)one_slot; }
    else if (T >= ncur) {        // no reduction: the output that stands for most reads wins, the first of them on ties (host loop: strictly greater replaces)
        unsigned long long best = 0ull;      // (cw + 1 capped to 40 bits) << 24 | (0xFFFFFF - order): larger weight first, then the earlier output
        uint32_t bslot = 0;
        for (uint32_t x = threadIdx.x; x < nj; x += 256) {
            const uint32_t j = j0 + x; const uint32_t c = Lv.out_n[j] & 0x7fffffffu; const uint32_t pos = Lv.job_pos[j];
            for (uint32_t sidx = 0; sidx < c; ++sidx) {
                unsigned long long cw = Lv.out_cw[(size_t)j * H.slots + sidx]; if (cw > 0xFFFFFFFFFEull) cw = 0xFFFFFFFFFEull;
                const unsigned long long key = ((cw + 1) << 24) | (unsigned long long)(0xFFFFFFu - ((pos + sidx) & 0xFFFFFFu));
                if (key > best) { best = key; bslot = j * (uint32_t)H.slots + sidx; }
            }
        }
        // block arg-max (keys are unique: the order is part of the key)
        unsigned long long kb = best; uint32_t sb = bslot;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const unsigned long long ok = __shfl_xor(kb, d); const uint32_t os = __shfl_xor(sb, d); if (ok > kb) { kb = ok; sb = os; } }
        if ((threadIdx.x & 63) == 0) { best_s[threadIdx.x >> 6] = kb; sm[4 + (threadIdx.x >> 6)] = sb; }
        __syncthreads();
        unsigned long long kk = 0ull; uint32_t ss = 0; for (int k = 0; k < 4; ++k) if (best_s[k] > kk) { kk = best_s[k]; ss = sm[4 + k]; }
        pick = (int)ss;
    }
    if (threadIdx.x == 0) {
        H.unit_pick[u] = pick;
        if (pick == -2) { H.unit_ncur[u] = T; tmp[0] = H.D > 0 ? poa_ntiles(T, (uint32_t)H.D) : 1u; tmp[1] = T; tmp[2] = 0; }
        else { tmp[0] = 0; tmp[1] = 0; tmp[2] = pick >= 0 ? (uint32_t)Lv.out_len[pick] : 0u; }
    }
}

// one workgroup: prefix sums over the units -> first tile / first sequence of every unit at the next level, arena offsets of the results
__global__ __launch_bounds__(1024) void k_poa_unit_offsets(HierDev H, int par)
{
    __shared__ uint32_t wsum[3][16]; __shared__ uint32_t carry[3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (H.ctrl[C_OVERFLOW]) { if (threadIdx.x == 0) H.ctrl[C_NJOBS + (par ^ 1)] = 0; return; }
    if (threadIdx.x < 3) carry[threadIdx.x] = threadIdx.x == 2 ? H.ctrl[C_RES_USED] : 0u;
    __syncthreads();
    for (uint32_t b = 0; b < H.U; b += 1024) {
        const uint32_t u = b + threadIdx.x; uint32_t v[3] = {0, 0, 0};
        if (u < H.U) { v[0] = H.unit_tmp[3ull * u]; v[1] = H.unit_tmp[3ull * u + 1]; v[2] = H.unit_tmp[3ull * u + 2]; }
        uint32_t x[3] = {v[0], v[1], v[2]};
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) for (int k = 0; k < 3; ++k) { const uint32_t y = __shfl_up(x[k], d); if (lane >= d) x[k] += y; }
        if (lane == 63) for (int k = 0; k < 3; ++k) wsum[k][w] = x[k];
        __syncthreads();
        uint32_t off[3], tot[3];
        for (int k = 0; k < 3; ++k) { off[k] = carry[k]; for (int q = 0; q < w; ++q) off[k] += wsum[k][q]; tot[k] = 0; for (int q = 0; q < 16; ++q) tot[k] += wsum[k][q]; }
        if (u < H.U) {
            H.unit_job0[par ^ 1][u] = off[0] + x[0] - v[0]; H.unit_njobs[par ^ 1][u] = v[0];
            H.unit_seq0[u] = off[1] + x[1] - v[1];
            if (H.unit_pick[u] >= 0) H.res_off[u] = off[2] + x[2] - v[2];
        }
        __syncthreads();
        if (threadIdx.x < 3) carry[threadIdx.x] += tot[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t nj = carry[0], ns = carry[1];
        H.unit_job0[par ^ 1][H.U] = nj;
        H.ctrl[C_RES_USED] = carry[2];
        bool ovf = nj > H.cap_jobs[par ^ 1] || ns > H.cap_jobs[par] * (uint32_t)H.slots || H.ctrl[C_OVERFLOW] != 0 || H.ctrl[C_FLAGS + 1] != 0 || H.ctrl[C_FLAGS + 2] != 0;
        if (ovf) H.ctrl[C_OVERFLOW] = 1;
        H.ctrl[C_NJOBS + (par ^ 1)] = ovf ? 0u : nj; H.ctrl[C_NSEQ] = ns;
    }
}

// one wave per tile of the level: its outputs become sequences of the next level (units that go on) or the unit's result (copied to the arena)
__global__ __launch_bounds__(256) void k_poa_fill_next(HierDev H, LevelDev Lv, int par)
{
    const int lane = threadIdx.x & 63;
    const uint32_t njobs = H.ctrl[C_NJOBS + par];
    if (H.ctrl[C_OVERFLOW]) return;
    for (uint32_t j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); j < njobs; j += gridDim.x * (blockDim.x >> 6)) {
    const uint32_t u = Lv.job_unit[j]; const int pick = H.unit_pick[u];
    const uint32_t n = Lv.out_n[j] & 0x7fffffffu;
    if (pick == -2) {
        if ((uint32_t)lane < n) {
            const size_t sl = (size_t)j * H.slots + lane; const uint64_t cw = Lv.out_cw[sl];
            PSeq S; S.s = Lv.out + sl * (size_t)H.capV; S.q = nullptr; S.len = Lv.out_len[sl];
            S.uw = cw > (1u << 20) ? (1 << 20) : (int)cw; if (S.uw < 1) S.uw = 1;
            S.cw = (uint32_t)(cw > 0xffffffffull ? 0xffffffffull : cw); S.mode = H.upper_mode; S.a0 = 0; S.a1 = -1;
            if (H.unit_bb[u] >= 0) {     // a tile consensus is a layer of the window like the reads it stands for: global only if it spans the window (oracle run_hierarchy)
                const int wlen = H.unit_wlen[u], offset = (int)(0.01 * (double)wlen), begin = Lv.out_span[2 * sl], end = Lv.out_span[2 * sl + 1];
                if (end >= begin) { S.a0 = begin; S.a1 = end; S.mode = (begin < offset && end > wlen - offset) ? NGSID_POA_GLOBAL : NGSID_POA_SEMI; }
            }
            if (S.len > H.Lmax) atomicExch(&H.ctrl[C_OVERFLOW], 1u);       // longer than the launch geometry was planned for: the host-driven loop takes over
            Lv.seqs[H.unit_seq0[u] + Lv.job_pos[j] + (uint32_t)lane] = S;
        }
    } else if (pick >= 0 && (uint32_t)pick / (uint32_t)H.slots == j) {
        const int len = Lv.out_len[pick]; const uint8_t* src = Lv.out + (size_t)pick * H.capV; uint8_t* dst = H.res + H.res_off[u];
        for (int x = lane; x < len; x += 64) dst[x] = src[x];
        if (H.want_cov) { const uint32_t* cs = Lv.out_cov + (size_t)pick * H.capV; uint32_t* cd = H.res_cov + H.res_off[u]; for (int x = lane; x < len; x += 64) cd[x] = cs[x]; }
        if (lane == 0) H.res_len[u] = len;
    }
    }
}

// tile lists of the next level: tile t of a unit takes its sequences [t D, (t + 1) D)
__global__ __launch_bounds__(256) void k_poa_fill_jobs(HierDev H, LevelDev Nx, int par /* of the level just done */)
{
    const uint32_t nj = H.ctrl[C_NJOBS + (par ^ 1)];
    if (H.ctrl[C_OVERFLOW]) return;
    const uint32_t* j0 = H.unit_job0[par ^ 1];
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nj; t += gridDim.x * blockDim.x) {
    uint32_t lo = 0, hi = H.U;                 // last unit whose first tile is <= t (units without tiles share the first tile of their successor and come before it)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (j0[mid] <= t) lo = mid; else hi = mid; }
    const uint32_t u = lo, tl = t - j0[u];
    const uint32_t T = H.unit_ncur[u]; const uint32_t Dl = H.D > 0 ? (uint32_t)H.D : T;
    Nx.job_off[t] = H.unit_seq0[u] + tl * Dl; Nx.job_bb[t] = H.unit_bb[u]; Nx.job_unit[t] = u;
    if (H.keep_final) Nx.job_final[t] = H.unit_njobs[par ^ 1][u] == 1 ? 1 : 0;            // (trim 3: the tile that ends its unit)
    if (t == nj - 1) Nx.job_off[nj] = H.ctrl[C_NSEQ];
    }
}


// returns NGSID_OK with *done = true when the device-driven hierarchy ran to completion; *done = false: nothing was changed, use the host-driven loop
int32_t run_hierarchy_dev(ngsid_ctx* ctx, const PSeq* d_level0, uint32_t maxlen0, const PSeq* d_bbs, const std::vector<int>& bb_len,
                          std::vector<Unit>& units, const HierParams& hp, bool* done)
{
    *done = false;
    static const bool want_ph = getenv("NGSID_POA_PHASES") != nullptr;       // the phase counters are read per level by the host-driven loop
    if (want_ph || ngsid_opt(ctx, "poa_host_levels", 0) || units.empty()) return NGSID_OK;
    const uint32_t U = (uint32_t)units.size();
    HostTimer ht(ctx->stream, "hierarchy (device levels)");
    // ---- level-0 tile lists (as in the host-driven loop) + per-unit records
    static thread_local PinVec<uint32_t> job_off, seq_idx, job_unit, u_u32; static thread_local PinVec<int32_t> job_bb, u_i32; static thread_local PinVec<uint8_t> job_final;
    job_off.clear(); job_off.push_back(0); seq_idx.clear(); job_unit.clear(); job_bb.clear(); job_final.clear();
    { size_t tot = 0; for (const Unit& Un : units) if (!Un.done) tot += Un.seqs.size(); seq_idx.reserve(tot); }
    u_u32.assign(3ull * U, 0); u_i32.assign(3ull * U, 0);        // ncur | job0 | njobs   and   bb | wlen | pick
    uint32_t maxD = 0; int maxbb = 0; bool any_nobb = false; uint64_t maxn = 0;
    for (uint32_t u = 0; u < U; ++u) {
        const Unit& Un = units[u];
        const uint32_t ncur = Un.done ? 0u : (uint32_t)Un.seqs.size();
        u_u32[u] = ncur; u_u32[U + u] = (uint32_t)job_unit.size(); u_i32[u] = Un.bb; u_i32[U + u] = Un.bb >= 0 ? bb_len[Un.bb] : 0; u_i32[2ull * U + u] = -2;
        if (ncur) {
            const uint32_t Dl = hp.D > 0 ? (uint32_t)hp.D : ncur;
            const uint32_t base = (uint32_t)seq_idx.size();
            seq_idx.insert(seq_idx.end(), Un.seqs.begin(), Un.seqs.end());          // the unit's sequences in one block copy: its tiles are consecutive slices of it
            const uint32_t nt = poa_ntiles(ncur, Dl);
            for (uint32_t t = 0; t < nt; ++t) {
                const uint32_t a = t * Dl, b = t + 1 == nt ? ncur : a + Dl;
                job_off.push_back(base + b); job_bb.push_back(Un.bb); job_unit.push_back(u); job_final.push_back(nt == 1 ? 1 : 0);
                maxD = std::max(maxD, b - a);
            }
            if (Un.bb >= 0) maxbb = std::max(maxbb, bb_len[Un.bb]); else any_nobb = true;
            maxn = std::max<uint64_t>(maxn, ncur);
        }
        u_u32[2ull * U + u] = (uint32_t)job_unit.size() - u_u32[U + u];
    }
    const uint32_t njobs0 = (uint32_t)job_unit.size();
    if (njobs0 == 0) return NGSID_OK;                       // (the host loop marks the empty units)
    // ---- one launch geometry for all levels: members of the upper levels are tile consensus sequences, planned for up to 5/4 of the longest input
    const int Lb = std::max<int>((int)maxlen0, maxbb), Lb2 = Lb + Lb / 4 + 16;
    const int maxL0 = std::max(maxbb, any_nobb ? Lb2 : 1);
    long long capV = (long long)maxL0 * (hp.node_cap > 0 ? hp.node_cap : 28) / 16; capV = std::max<long long>(capV, maxL0 + 64); capV = std::max<long long>(capV, (long long)Lb2 + 1);
    capV = (capV + 7) & ~7ll;
    const int Lmax = (int)(((uint32_t)std::max(Lb2, 1) + 15) & ~15u);
    const int band0 = hp.band <= 64 ? 64 : (hp.band <= 128 ? 128 : 256);
    if (capV > 0xFFF0 || 3 * capV / 2 > 0xFFF0 || (long long)hp.m * Lmax >= 65536 || hp.m < 0 || hp.g >= 0) return NGSID_OK;      // the host loop reports what is wrong (or fits where this margin does not)
    for (int bw = band0; bw <= 256; bw *= 2) if (poa_lds_bytes((int)capV, (int)(3 * capV / 2), Lmax, bw) > 160 * 1024) return NGSID_OK;
    const int slots = (int)std::min<uint32_t>(maxD, (uint32_t)std::max<long long>(1, ngsid_opt(ctx, "poa_out_slots", 4)));      // output slots per tile of the first attempt (a tile that closes its graph more often falls back to the host-driven loop, which retries with more)
    // levels the LARGEST unit needs when every tile returns one consensus (the tile counts of poa_ntiles, level by level).  Round 6: exactly that many are enqueued before the first
    // synchronisation (it used to be two more - an off-by-one and a spare - i.e. 2 x 9 empty dispatches per hierarchy); a tile that closes its graph early and returns two
    // sequences can make a unit need one level more: the loop below then enqueues three more after looking at the counters.
    int levels_est = 0; { uint64_t n = maxn; const uint32_t Dd = hp.D > 0 ? (uint32_t)hp.D : (uint32_t)std::min<uint64_t>(maxn, 0xffffffffull); while (n > 1 && levels_est < C_MAXLV - 4) { n = poa_ntiles((uint32_t)n, std::max<uint32_t>(Dd, 2u)); ++levels_est; } }
    levels_est = std::min(std::max(levels_est, 1), C_MAXLV - 2);
    uint32_t cap[2]; cap[0] = njobs0;
    { const uint64_t e1 = hp.D > 0 ? (njobs0 + (uint64_t)hp.D - 1) / (uint64_t)hp.D * (uint64_t)slots + U : U; cap[1] = (uint32_t)std::min<uint64_t>(njobs0, e1 + e1 / 4 + 1024); }
    PoaPlan plan{(int)capV, (int)(3 * capV / 2), Lmax, 0, 0, band0};
    { int32_t rc = poa_prepare(ctx, plan, njobs0); if (rc) return rc; }
    // ---- buffers
    const bool need_cov = hp.want_cov || hp.trim_tiles;
    LevelDev L[2];
    for (int q = 0; q < 2; ++q) {
        ngsid_ctx::PoaLevelBufs* B = &ctx->poa_lv[q]; const size_t ns = (size_t)cap[q] * slots;
        HIPCHK(ctx, B->out.reserve(ns * capV)); HIPCHK(ctx, B->out_len.reserve(ns)); HIPCHK(ctx, B->out_cw.reserve(ns)); HIPCHK(ctx, B->out_span.reserve(ns * 2)); HIPCHK(ctx, B->out_n.reserve(cap[q]));
        if (need_cov) HIPCHK(ctx, B->out_cov.reserve(ns * capV));
        HIPCHK(ctx, B->seqs.reserve(sizeof(PSeq) * ns)); HIPCHK(ctx, B->job_off.reserve((size_t)cap[q] + 1)); HIPCHK(ctx, B->job_bb.reserve(cap[q])); HIPCHK(ctx, B->job_list.reserve(cap[q]));
        HIPCHK(ctx, B->job_unit.reserve(cap[q])); HIPCHK(ctx, B->job_pos.reserve(cap[q])); HIPCHK(ctx, B->job_final.reserve(cap[q]));
        L[q] = LevelDev{(PSeq*)B->seqs.p, B->out.p, B->out_len.p, B->out_span.p, B->out_cw.p, B->out_n.p, need_cov ? B->out_cov.p : nullptr, B->job_off.p, B->job_bb.p, B->job_unit.p, B->job_pos.p, B->job_list.p, B->job_final.p};
    }
    HIPCHK(ctx, ctx->poa_lv[0].seq_idx.reserve(seq_idx.size()));
    DevBuf<uint32_t> d_u32, d_ctrl, d_res_off, d_res_cov; DevBuf<int32_t> d_i32, d_res_len; DevBuf<uint8_t> d_res;
    HIPCHK(ctx, d_u32.alloc(9ull * U + 4)); HIPCHK(ctx, d_i32.alloc(3ull * U)); HIPCHK(ctx, d_ctrl.alloc(C_WORDS)); HIPCHK(ctx, d_res_off.alloc(U)); HIPCHK(ctx, d_res_len.alloc(U));
    HIPCHK(ctx, d_res.alloc((size_t)U * capV)); if (hp.want_cov) HIPCHK(ctx, d_res_cov.alloc((size_t)U * capV));
    HierDev H{};
    H.U = U; H.D = hp.D; H.slots = slots; H.capV = (int)capV; H.upper_mode = hp.upper_mode; H.want_cov = hp.want_cov ? 1 : 0; H.Lmax = Lmax; H.keep_final = (hp.trim_tiles & 4) ? 1 : 0; H.cap_jobs[0] = cap[0]; H.cap_jobs[1] = cap[1];
    H.unit_bb = d_i32.p; H.unit_wlen = d_i32.p + U; H.unit_pick = d_i32.p + 2ull * U;
    H.unit_ncur = d_u32.p; H.unit_job0[0] = d_u32.p + U; H.unit_njobs[0] = d_u32.p + 2ull * U + 1; H.unit_job0[1] = d_u32.p + 3ull * U + 1; H.unit_njobs[1] = d_u32.p + 4ull * U + 2; H.unit_seq0 = d_u32.p + 5ull * U + 2; H.unit_tmp = d_u32.p + 6ull * U + 2;
    H.res_off = d_res_off.p; H.res_len = d_res_len.p; H.res = d_res.p; H.res_cov = hp.want_cov ? d_res_cov.p : nullptr; H.ctrl = d_ctrl.p;
    HIPCHK(ctx, hipMemsetAsync(d_ctrl.p, 0, sizeof(uint32_t) * C_WORDS, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(d_res_len.p, 0xff, sizeof(int32_t) * U, ctx->stream));                                    // -1 = no result
    HIPCHK(ctx, hipMemcpyAsync(H.unit_ncur, u_u32.data(), 4ull * U, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(H.unit_job0[0], u_u32.data() + U, 4ull * U, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(H.unit_njobs[0], u_u32.data() + 2ull * U, 4ull * U, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_i32.p, u_i32.data(), 12ull * U, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(L[0].job_off, job_off.data(), 4 * job_off.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->poa_lv[0].seq_idx.p, seq_idx.data(), 4 * seq_idx.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(L[0].job_bb, job_bb.data(), 4 * job_bb.size(), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(L[0].job_unit, job_unit.data(), 4 * job_unit.size(), hipMemcpyHostToDevice, ctx->stream));
    if (hp.trim_tiles & 4) HIPCHK(ctx, hipMemcpyAsync(L[0].job_final, job_final.data(), job_final.size(), hipMemcpyHostToDevice, ctx->stream));
    static thread_local PinVec<uint32_t> h_nj; h_nj.assign(1, njobs0);
    HIPCHK(ctx, hipMemcpyAsync(d_ctrl.p + C_NJOBS, h_nj.data(), 4, hipMemcpyHostToDevice, ctx->stream));
    ht.mark("L0 lists + upload");
    static thread_local PinVec<uint32_t> h_ctrl; h_ctrl.resize(C_WORDS);
    int level = 0;
    for (;;) {
        const int batch_end = std::min(C_MAXLV - 1, level + (level == 0 ? levels_est : 3));
        for (; level < batch_end; ++level) {
            const int par = level & 1; LevelDev& Lv = L[par]; LevelDev& Nx = L[par ^ 1];
            PoaJobSet J{};
            J.seqs = level == 0 ? d_level0 : (const PSeq*)Nx.seqs; J.bbs = d_bbs; J.seq_idx = level == 0 ? ctx->poa_lv[0].seq_idx.p : nullptr; J.job_off = Lv.job_off; J.job_bb = Lv.job_bb; J.njobs = cap[par];
            J.m = hp.m; J.n = hp.n; J.g = hp.g; J.Vcap = (int)capV; J.Ecap = (int)(3 * capV / 2); J.Lmax = Lmax; J.D = slots; J.node_cap = hp.node_cap; J.trim_tiles = (hp.trim_tiles & 1) | ((level > 0 && (hp.trim_tiles & 1)) ? 2 : 0); J.job_final = (hp.trim_tiles & 4) ? Lv.job_final : nullptr;
            J.out = Lv.out; J.out_len = Lv.out_len; J.out_cw = Lv.out_cw; J.out_n = Lv.out_n; J.out_cov = Lv.out_cov; J.out_span = Lv.out_span;
            J.dropped = d_ctrl.p + C_FLAGS; J.slot_overflow = d_ctrl.p + C_FLAGS + 1;
            J.job_list = nullptr; J.nrun = 0; J.nrun_dev = d_ctrl.p + C_NJOBS + par;
            int32_t rc = poa_launch(ctx, plan, J, band0, false, d_ctrl.p + C_WORK + 3 * level); if (rc) return rc;
            int stage = 0;
            for (int bw = band0 * 2; bw <= 256; bw *= 2, ++stage) {          // band-edge check (oracle run_tile): flagged tiles run again, whole, with twice the band
                uint32_t* cnt = d_ctrl.p + C_REDO + 2 * level + stage;
                hipLaunchKernelGGL(k_poa_redo_list, dim3(std::min<uint32_t>((cap[par] + 255) / 256, 1024u)), dim3(256), 0, ctx->stream, Lv.out_n, d_ctrl.p + C_NJOBS + par, Lv.job_list, cnt, d_ctrl.p + C_REDO_TOTAL);
                PoaJobSet R = J; R.job_list = Lv.job_list; R.nrun_dev = cnt;
                rc = poa_launch(ctx, plan, R, bw, true, d_ctrl.p + C_WORK + 3 * level + 1 + stage); if (rc) return rc;
            }
            hipLaunchKernelGGL(k_poa_unit_scan, dim3(U), dim3(256), 0, ctx->stream, H, Lv, par);
            hipLaunchKernelGGL(k_poa_unit_offsets, dim3(1), dim3(1024), 0, ctx->stream, H, par);
            hipLaunchKernelGGL(k_poa_fill_next, dim3(std::min<uint32_t>((cap[par] + 3) / 4, 4096u)), dim3(256), 0, ctx->stream, H, Lv, par);
            hipLaunchKernelGGL(k_poa_fill_jobs, dim3(std::min<uint32_t>((cap[par ^ 1] + 255) / 256, 1024u)), dim3(256), 0, ctx->stream, H, Nx, par);
            HIPCHK(ctx, hipGetLastError());
        }
        HIPCHK(ctx, hipMemcpyAsync(h_ctrl.data(), d_ctrl.p, 4 * C_WORDS, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (h_ctrl[C_FLAGS + 2]) NGSID_FAIL(ctx, NGSID_ERR_HIP, "internal: POA tile kernel loop guard tripped (code %u)", h_ctrl[C_FLAGS + 2]);
        if (h_ctrl[C_OVERFLOW] || h_ctrl[C_FLAGS + 1]) { ht.mark("gave up (geometry): host-driven loop"); return NGSID_OK; }
        if (h_ctrl[C_NJOBS + (level & 1)] == 0) break;                        // every unit has finished
        if (level >= C_MAXLV - 1) return NGSID_OK;
    }
    ctx->poa_redo_tiles += h_ctrl[C_REDO_TOTAL];
    ht.mark("levels");
    // ---- results
    static thread_local PinVec<int32_t> h_len; static thread_local PinVec<uint32_t> h_off, h_cov; static thread_local PinVec<uint8_t> h_res;
    h_len.resize(U); h_off.resize(U); const size_t used = h_ctrl[C_RES_USED];
    HIPCHK(ctx, hipMemcpyAsync(h_len.data(), d_res_len.p, 4ull * U, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_off.data(), d_res_off.p, 4ull * U, hipMemcpyDeviceToHost, ctx->stream));
    h_res.resize(used + 1); if (used) HIPCHK(ctx, hipMemcpyAsync(h_res.data(), d_res.p, used, hipMemcpyDeviceToHost, ctx->stream));
    if (hp.want_cov) { h_cov.resize(used + 1); if (used) HIPCHK(ctx, hipMemcpyAsync(h_cov.data(), d_res_cov.p, 4 * used, hipMemcpyDeviceToHost, ctx->stream)); }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t u = 0; u < U; ++u) {
        Unit& Un = units[u]; if (Un.done) continue;
        Un.done = true; Un.seqs.clear();
        if (h_len[u] < 0) continue;
        const int len = h_len[u];
        Un.result.assign((const char*)h_res.data() + h_off[u], (size_t)len);
        if (hp.want_cov) Un.cov.assign(h_cov.data() + h_off[u], h_cov.data() + h_off[u] + len);
        Un.has_result = true;
    }
    ht.mark("results");
    *done = true;
    return NGSID_OK;
}

static int32_t run_hierarchy_batch(ngsid_ctx* ctx, const PSeq* d_level0, uint32_t maxlen0, const PSeq* d_bbs, const std::vector<int>& bb_len,
                                   std::vector<Unit>& units, const HierParams& hp)
{
    bool done = false;
    int32_t rc = run_hierarchy_dev(ctx, d_level0, maxlen0, d_bbs, bb_len, units, hp, &done);
    if (rc || done) return rc;
    return run_hierarchy_host(ctx, d_level0, maxlen0, d_bbs, bb_len, units, hp);
}

// Round 5: the level buffers of a hierarchy are sized by its level-0 tiles (output slots x graph capacity bytes per tile, + 4 bytes per base of
// coverage when tiles are trimmed: 17.6 KB per tile of a 500-base polishing window), i.e. by the number of READS of the call - 107 GB for the two windows of 10 M
// 750-base reads.  Units (clusters / windows) are independent hierarchies, so the call is cut into batches of whole units under a byte budget the context derives
// from the HBM that is free NOW (a third of it, divided by the live contexts of the process, at most 48 GB, at least 1 GB; "poa_level_budget_mb" sets it for tests).  Results cannot depend on the batching: a
// unit's hierarchy never sees another unit, and the launch geometry a batch plans for (graph capacity, slots) only decides whether the device-driven levels or
// the host-driven loop run it (tests run one batch per unit against one batch for all).  A single unit larger than the budget still runs alone: the footprint is
// bounded by max(budget, largest unit), not by the call.
int32_t run_hierarchy(ngsid_ctx* ctx, const PSeq* d_level0, uint32_t maxlen0, const PSeq* d_bbs, const std::vector<int>& bb_len,
                      std::vector<Unit>& units, const HierParams& hp)
{
    if (hp.single_below > 0) {
        // round 6 (ngsid_poa_params_t.single_below, oracle run_unit): the units the caller marked (fewer sequences than the threshold) run as ONE graph in the given order, as
        // hierarchies of their own - with room for NGSID_POA_SINGLE_NODE_CAP / 16 times the first sequence where the unit's longest sequence has at most NGSID_POA_SINGLE_MAXLEN
        // bases (launch geometry for that class only), with the caller's capacity otherwise; the others are tiled at depth D exactly as before.  A call without such units
        // takes the unchanged path.
        HierParams hb = hp; hb.single_below = 0;
        for (int cls = 0; cls < 2; ++cls) {
            std::vector<size_t> pick; uint32_t mx = 0;
            for (size_t u = 0; u < units.size(); ++u) if (!units[u].done && units[u].single_maxlen > 0 && (units[u].single_maxlen <= NGSID_POA_SINGLE_MAXLEN) == (cls == 0)) { pick.push_back(u); mx = std::max(mx, units[u].single_maxlen); }
            if (pick.empty()) continue;
            HierParams hs = hb; hs.D = 0; if (cls == 0) hs.node_cap = NGSID_POA_SINGLE_NODE_CAP;
            std::vector<Unit> sub; sub.reserve(pick.size());
            for (size_t u : pick) sub.push_back(std::move(units[u]));
            const int32_t rc = run_hierarchy(ctx, d_level0, std::min(maxlen0, mx), d_bbs, bb_len, sub, hs);
            for (size_t x = 0; x < pick.size(); ++x) { units[pick[x]] = std::move(sub[x]); units[pick[x]].done = true; }
            if (rc) return rc;
        }
        bool any = false; for (const Unit& U : units) any = any || !U.done;
        if (!any) return NGSID_OK;
        return run_hierarchy(ctx, d_level0, maxlen0, d_bbs, bb_len, units, hb);      // the large units (the others are done: every loop below skips them)
    }
    size_t budget = 0;
    { const long long mb = ngsid_opt(ctx, "poa_level_budget_mb", 0);
      if (mb > 0) budget = (size_t)mb << 20;
      else {
          size_t freeb = 0, totalb = 0; if (hipMemGetInfo(&freeb, &totalb) != hipSuccess) freeb = (size_t)16 << 30;
          size_t own = 0; for (auto& L : ctx->poa_lv) own += L.out.abytes + L.out_cov.abytes + L.seqs.abytes;       // grow-only buffers of this context that the call will reuse
          budget = std::min<size_t>(std::max<size_t>((freeb + own + ngsid_pool_cached_bytes()) / (3 * (size_t)ngsid_pool_contexts()), (size_t)1 << 30), (size_t)48 << 30);      // (contexts of one process share the device: eight virtual ranks each take a 24th)
      } }
    // bytes per level-0 tile: both ping-pong level buffers (the second holds ~slots / D of the first + 25 %), as run_hierarchy_dev sizes them
    int maxbb = 0; for (const Unit& U : units) if (!U.done && U.bb >= 0) maxbb = std::max(maxbb, bb_len[U.bb]);
    const int Lb = std::max<int>((int)maxlen0, maxbb), Lb2 = Lb + Lb / 4 + 16;
    long long capV = (long long)Lb2 * (hp.node_cap > 0 ? hp.node_cap : 28) / 16; capV = std::max<long long>(capV, Lb2 + 64);
    const long long slots = std::max<long long>(1, ngsid_opt(ctx, "poa_out_slots", 4));
    const bool need_cov = hp.want_cov || hp.trim_tiles;
    const double per_tile = (double)slots * ((double)capV * (need_cov ? 5.0 : 1.0) + 64.0) * (hp.D > 0 ? 1.0 + 1.25 * (double)slots / (double)hp.D : 1.0) + 32.0;
    // batches of consecutive units
    std::vector<std::pair<size_t, size_t>> batches; size_t b0 = 0; double acc = 0;
    for (size_t u = 0; u < units.size(); ++u) {
        const uint32_t ncur = units[u].done ? 0u : (uint32_t)units[u].seqs.size();
        const double need = ncur ? per_tile * (double)poa_ntiles(ncur, hp.D > 0 ? (uint32_t)hp.D : ncur) : 0.0;
        if (u > b0 && acc + need > (double)budget) { batches.push_back({b0, u}); b0 = u; acc = 0; }
        acc += need;
    }
    batches.push_back({b0, units.size()});
    if (batches.size() == 1) return run_hierarchy_batch(ctx, d_level0, maxlen0, d_bbs, bb_len, units, hp);
    for (auto& bt : batches) {
        std::vector<Unit> sub; sub.reserve(bt.second - bt.first);
        for (size_t u = bt.first; u < bt.second; ++u) sub.push_back(std::move(units[u]));
        const int32_t rc = run_hierarchy_batch(ctx, d_level0, maxlen0, d_bbs, bb_len, sub, hp);
        for (size_t u = bt.first; u < bt.second; ++u) units[u] = std::move(sub[u - bt.first]);
        if (rc) return rc;
    }
    return NGSID_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- (a13,a14)
static int32_t poa_consensus_impl(ngsid_ctx* ctx, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                  const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed, uint32_t* cov, const uint32_t* weight = nullptr);

// (8e step 3) the merge of per-shard partial consensuses: sequence i stands for weight[i] reads (host array, one entry per read of `reads`; 0 counts as 1).
// Qualities are ignored: every base of sequence i weighs weight[i] (capped at 2^20 per base like the tile consensuses of the hierarchy's upper levels).
extern "C" int32_t ngsid_poa_consensus_weighted(ngsid_ctx* ctx, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                                const ngsid_poa_params_t* prm, const uint32_t* weight, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed)
{
    if (ctx && !weight) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null weight array");
    return poa_consensus_impl(ctx, reads, read_order, grp_off, n_groups, prm, cons_off, cons, cons_cap, needed, nullptr, weight);
}

extern "C" int32_t ngsid_poa_consensus(ngsid_ctx* ctx, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                       const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed)
{
    ApiClock api_clock_(ctx, "poa_consensus");
    return poa_consensus_impl(ctx, reads, read_order, grp_off, n_groups, prm, cons_off, cons, cons_cap, needed, nullptr);
}

// same, plus the per-base coverage of the consensus (boundary 8b: "consensus bytes, len, per-base coverage")
extern "C" int32_t ngsid_poa_consensus_cov(ngsid_ctx* ctx, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                           const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint32_t* cov, uint64_t cons_cap, uint64_t* needed)
{
    if (ctx && !cov) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null coverage buffer");
    return poa_consensus_impl(ctx, reads, read_order, grp_off, n_groups, prm, cons_off, cons, cons_cap, needed, cov);
}

static int32_t poa_consensus_impl(ngsid_ctx* ctx, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                  const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed, uint32_t* cov, const uint32_t* weight)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!reads || !grp_off || !prm || !cons_off) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    HostTimer htc(ctx->stream, "consensus");
    DevReads RD; int32_t rc = ngsid_upload_reads(ctx, reads, &RD, false); if (rc) return rc;
    htc.mark("upload");
    if (!read_order && grp_off[n_groups] > RD.n) NGSID_FAIL(ctx, NGSID_ERR_ARG, "group offsets exceed the read set");
    if (read_order) for (uint64_t x = 0; x < grp_off[n_groups]; ++x) if (read_order[x] >= RD.n) NGSID_FAIL(ctx, NGSID_ERR_ARG, "read_order[%llu] out of range", (unsigned long long)x);
    htc.mark("checks");
    DevBuf<PSeq> d_seqs; HIPCHK(ctx, d_seqs.alloc(RD.n));
    DevBuf<uint32_t> d_weight; if (weight && RD.n) { HIPCHK(ctx, d_weight.alloc(RD.n)); HIPCHK(ctx, hipMemcpyAsync(d_weight.p, weight, 4 * RD.n, hipMemcpyHostToDevice, ctx->stream)); }
    if (RD.n) hipLaunchKernelGGL(k_make_pseq_reads, dim3((unsigned)((RD.n + 255) / 256)), dim3(256), 0, ctx->stream, RD.seq, RD.qual, RD.off, RD.n, prm->mode, weight ? d_weight.p : (const uint32_t*)nullptr, d_seqs.p);
    HIPCHK(ctx, hipGetLastError());
    std::vector<Unit> units(n_groups);
    for (uint64_t g = 0; g < n_groups; ++g) {
        if (read_order) units[g].seqs.assign(read_order + grp_off[g], read_order + grp_off[g + 1]);
        else { units[g].seqs.resize(grp_off[g + 1] - grp_off[g]); for (uint64_t r = grp_off[g]; r < grp_off[g + 1]; ++r) units[g].seqs[r - grp_off[g]] = (uint32_t)r; }
    }
    HierParams hp{prm->match, prm->mismatch, prm->gap, prm->band > 0 ? prm->band : (RD.maxlen <= NGSID_POA_BAND64_MAXLEN ? 64 : 128), prm->node_cap, prm->tile_depth, prm->mode, cov != nullptr, prm->trim > 0 ? 1 : 0};
    hp.single_below = prm->single_below > 0 ? prm->single_below : 0;
    if (hp.single_below > 0) for (Unit& U : units) if (!U.seqs.empty() && U.seqs.size() < (size_t)hp.single_below) { uint32_t mx = 1; for (uint32_t r : U.seqs) mx = std::max<uint32_t>(mx, (uint32_t)(RD.h_off[r + 1] - RD.h_off[r])); U.single_maxlen = mx; }
    std::vector<int> nobb;
    htc.mark("units");
    rc = run_hierarchy(ctx, d_seqs.p, RD.maxlen, nullptr, nobb, units, hp); if (rc) return rc;
    htc.mark("hierarchy");
    uint64_t total = 0; bool overflow = false; cons_off[0] = 0;
    for (uint64_t g = 0; g < n_groups; ++g) {
        const std::string& s = units[g].result;
        if (total + s.size() <= cons_cap && cons) { memcpy(cons + total, s.data(), s.size()); if (cov && units[g].cov.size() == s.size() && !s.empty()) memcpy(cov + total, units[g].cov.data(), 4 * s.size()); else if (cov) for (size_t x = 0; x < s.size(); ++x) cov[total + x] = 0; }
        else if (s.size()) overflow = true;
        total += s.size(); cons_off[g + 1] = total;
    }
    if (needed) *needed = total;
    if (overflow) NGSID_FAIL(ctx, NGSID_ERR_CAPACITY, "consensus buffer too small: need %llu bytes", (unsigned long long)total);
    return NGSID_OK;
}

// ---------------------------------------------------------------------------------------------- (a16,a17)
namespace {

__global__ __launch_bounds__(256)
void k_strand(const uint64_t* __restrict__ mzoff /* compact CSR offsets of the reads' minimizers */, const uint32_t* __restrict__ mzcnt, const uint32_t* __restrict__ hlen, const uint64_t* __restrict__ mzcode, int k,
              const uint32_t* __restrict__ rgroup, const uint64_t* __restrict__ bcodes, const uint64_t* __restrict__ boff /* 2G+1: fw lists then rc lists */, uint32_t G,
              uint64_t n, uint8_t* __restrict__ orient)
{
    const int lane = threadIdx.x & 63;
    const uint64_t r = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= n) return;
    const uint32_t g = rgroup[r];
    if (g == 0xffffffffu) { if (lane == 0) orient[r] = 255; return; }        // read belongs to no group
    const uint32_t M = hlen[r] >= (uint32_t)k ? mzcnt[r] : 0;
    const uint64_t* cf = bcodes + boff[g]; const uint32_t nf = (uint32_t)(boff[g + 1] - boff[g]);
    const uint64_t* cr = bcodes + boff[G + g]; const uint32_t nr = (uint32_t)(boff[G + g + 1] - boff[G + g]);
    int a = 0, b = 0;
    for (uint32_t x = lane; x < M; x += 64) {
        const uint64_t code = mzcode[mzoff[r] + x];
        uint32_t lo = 0, hi = nf; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cf[mid] < code) lo = mid + 1; else hi = mid; }
        a += (lo < nf && cf[lo] == code);
        lo = 0; hi = nr; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cr[mid] < code) lo = mid + 1; else hi = mid; }
        b += (lo < nr && cr[lo] == code);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
    if (lane == 0) orient[r] = (a == 0 && b == 0) ? 255 : (b > a ? 1 : 0);
}

// Polishing windows of a backbone of Bl bases: W bases each; a last window shorter than W/10 is merged into the one before it.  (racon keeps it:
// only layers of at least 0.02 W bases enter a window, so a 7-base tail window is built from the reads that carry INSERTIONS there and the
// consensus gains bases - seen on 5 007-base amplicons.)  Window w covers [w * W, w == nwin - 1 ? Bl : (w + 1) * W).
__host__ __device__ __forceinline__ int polish_nwin(int Bl, int W) { const int raw = Bl <= W ? 1 : (Bl + W - 1) / W; const int tail = Bl - (raw - 1) * W; return (raw >= 2 && tail < W / 10) ? raw - 1 : raw; }
__host__ __device__ __forceinline__ int polish_wlen(int Bl, int W, int w) { return w == polish_nwin(Bl, W) - 1 ? Bl - w * W : W; }

__device__ __forceinline__ uint8_t comp_base(uint8_t c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

__global__ void k_orient(const uint8_t* __restrict__ seq, const uint8_t* __restrict__ qual, const uint64_t* __restrict__ off, uint64_t n,
                         const uint8_t* __restrict__ orient, uint8_t* __restrict__ oseq, uint8_t* __restrict__ oqual)
{
    const uint64_t r = blockIdx.x;
    if (r >= n) return;
    const uint64_t b = off[r]; const int len = (int)(off[r + 1] - b); const bool rc = orient[r] == 1;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        if (rc) { oseq[b + i] = comp_base(seq[b + len - 1 - i]); if (qual) oqual[b + i] = qual[b + len - 1 - i]; }
        else { oseq[b + i] = seq[b + i]; if (qual) oqual[b + i] = qual[b + i]; }
    }
}

// one wave per (pair, window): validity filters of racon's window assignment + the PSeq of the layer (oracle ongsid_polish);
// the lanes share the mean-quality sum over the layer's bases (coalesced), everything else is wave-uniform
__global__ __launch_bounds__(256) void k_layers(const uint8_t* __restrict__ oseq, const uint8_t* __restrict__ oqual, const uint64_t* __restrict__ off,
                         const uint32_t* __restrict__ pair_read, const uint32_t* __restrict__ pair_group, uint64_t npairs, int nwinmax,
                         const int32_t* __restrict__ bp, const int32_t* __restrict__ span, const int32_t* __restrict__ blen /* per group */,
                         int W, double qthr, double ethr, PSeq* __restrict__ lay, uint16_t* __restrict__ valid /* 0 = no layer, else 1 + its first window position */, int* __restrict__ maxlen_out)
{
    const int lane = threadIdx.x & 63;
    const uint64_t t = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= npairs * (uint64_t)nwinmax) return;
    const uint64_t p = t / nwinmax; const int wdx = (int)(t % nwinmax);
    if (lane == 0) valid[t] = 0;
    const int qb = span[p * 4 + 0], qe = span[p * 4 + 1], tb = span[p * 4 + 2], te = span[p * 4 + 3];
    if (qb < 0) return;
    const int qs = qe - qb + 1, ts = te - tb + 1; const int mn = qs < ts ? qs : ts, mx = qs < ts ? ts : qs;
    if (1.0 - (double)mn / (double)mx > ethr) return;
    const int Bl = blen[pair_group[p]]; const int nw = polish_nwin(Bl, W);
    if (wdx >= nw) return;
    const int32_t* b = bp + (p * (uint64_t)nwinmax + wdx) * 4;
    int qf = b[0], ql = b[1], tf = b[2], tl = b[3];
    if (wdx == nw - 1 && wdx + 1 < nwinmax && (Bl + W - 1) / W > nw) {       // merged tail window: the aligner's break points of the two raw windows are joined
        const int32_t* b2 = b + 4;
        if (b2[0] >= 0) { if (qf < 0) { qf = b2[0]; tf = b2[2]; } ql = b2[1]; tl = b2[3]; }
    }
    if (qf < 0) return;
    const int len = ql - qf + 1; if ((double)len < 0.02 * (double)W) return;
    const uint32_t read = pair_read[p]; const uint64_t rb = off[read];
    if (oqual) {
        long long sq = 0;
        for (int x = qf + lane; x <= ql; x += 64) sq += (long long)oqual[rb + x] - 33;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sq += __shfl_xor(sq, d);
        if ((double)sq / (double)len < qthr) return;
    }
    const int ws = wdx * W; const int wlen = polish_wlen(Bl, W, wdx);
    const int begin = tf - ws, end = tl - ws; const int offset = (int)(0.01 * (double)wlen);
    PSeq S; S.s = oseq + rb + qf; S.q = oqual ? oqual + rb + qf : nullptr; S.len = len; S.uw = 1; S.cw = 1; S.a0 = begin; S.a1 = end;
    S.mode = (begin < offset && end > wlen - offset) ? NGSID_POA_GLOBAL : NGSID_POA_SEMI;
    if (lane == 0) { lay[t] = S; valid[t] = (uint16_t)((begin < 0 ? 0 : (begin > 65533 ? 65533 : begin)) + 1); if (len > __atomic_load_n(maxlen_out, __ATOMIC_RELAXED)) atomicMax(maxlen_out, len); }      // (millions of atomics on one word would serialise)
}

std::string revcomp(const std::string& s) { std::string r(s.size(), 'N'); for (size_t i = 0; i < s.size(); ++i) { char c = s[s.size() - 1 - i]; r[i] = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N'; } return r; }

}  // namespace

struct PolishTrace { std::vector<std::string> seq; std::vector<uint64_t> used;           // [it * G + g]: the backbones after every iteration
                     int32_t* aln = nullptr; };                                          // ngsid_polish_trace_aln: [(it * n_listed + x) * 6 ...] = the read -> backbone alignment of listed read x in iteration it

static int32_t polish_impl(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                           const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                           uint64_t* out_off, uint8_t* out, uint64_t out_cap, uint64_t* needed, uint64_t* n_used, PolishTrace* trace);

extern "C" int32_t ngsid_polish(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                                const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                                uint64_t* out_off, uint8_t* out, uint64_t out_cap, uint64_t* needed, uint64_t* n_used)
{
    ApiClock api_clock_(ctx, "polish");
    return polish_impl(ctx, backbones, reads, read_order, grp_off, n_groups, prm, out_off, out, out_cap, needed, n_used, nullptr);
}

static int32_t polish_trace_impl(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                                 const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                                 uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used, int32_t* it_aln)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!prm || !it_off || prm->iters < 1) NGSID_FAIL(ctx, NGSID_ERR_ARG, "ngsid_polish_trace: null argument or iters < 1");
    PolishTrace tr; tr.aln = it_aln; std::vector<uint64_t> ooff(n_groups + 1, 0); uint64_t need1 = 0;
    const int32_t rc = polish_impl(ctx, backbones, reads, read_order, grp_off, n_groups, prm, ooff.data(), nullptr, 0, &need1, nullptr, &tr);
    if (rc != NGSID_OK) return rc;                                        // (with a trace the inner call copies nothing out and checks no capacity: ADVICE r4)
    if (tr.seq.size() != (size_t)prm->iters * n_groups || tr.used.size() != tr.seq.size()) NGSID_FAIL(ctx, NGSID_ERR_HIP, "internal: polish trace holds %zu entries for %d x %llu", tr.seq.size(), (int)prm->iters, (unsigned long long)n_groups);
    uint64_t total = 0; bool ovf = false; it_off[0] = 0;
    for (size_t x = 0; x < tr.seq.size(); ++x) {
        if (it_out && total + tr.seq[x].size() <= it_cap) memcpy(it_out + total, tr.seq[x].data(), tr.seq[x].size()); else if (tr.seq[x].size()) ovf = true;
        total += tr.seq[x].size(); it_off[x + 1] = total; if (it_used) it_used[x] = tr.used[x];
    }
    if (needed) *needed = total;
    if (ovf) NGSID_FAIL(ctx, NGSID_ERR_CAPACITY, "trace buffer too small: need %llu bytes", (unsigned long long)total);
    return NGSID_OK;
}
extern "C" int32_t ngsid_polish_trace(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                                      const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                                      uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used)
{
    ApiClock api_clock_(ctx, "polish_trace");
    return polish_trace_impl(ctx, backbones, reads, read_order, grp_off, n_groups, prm, it_off, it_out, it_cap, needed, it_used, nullptr);
}
// (a17, boundary 8b(1)) ngsid_polish_trace + what minimap2 leaves in read_alignments_it_{i}.paf (consensus.py:112-121): the read -> backbone alignment of every LISTED read in
// every iteration, it_aln[(it * n_listed + x) * 6 ...] = {strand (0 +, 1 -, -1 = not aligned: no PAF line), q_begin, q_end, t_begin, t_end, edit distance}, PAF coordinates
// (0-based, end exclusive, the query interval on the read's ORIGINAL strand); n_listed = grp_off[n_groups].
extern "C" int32_t ngsid_polish_trace_aln(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                                          const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                                          uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used, int32_t* it_aln)
{
    ApiClock api_clock_(ctx, "polish_trace");
    if (ctx && !it_aln) NGSID_FAIL(ctx, NGSID_ERR_ARG, "ngsid_polish_trace_aln: null alignment buffer");
    return polish_trace_impl(ctx, backbones, reads, read_order, grp_off, n_groups, prm, it_off, it_out, it_cap, needed, it_used, it_aln);
}

static int32_t polish_impl(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                           const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                           uint64_t* out_off, uint8_t* out, uint64_t out_cap, uint64_t* needed, uint64_t* n_used, PolishTrace* trace)
{
    if (!ctx) return NGSID_ERR_ARG;
    if (!backbones || !reads || !grp_off || !prm || !out_off) NGSID_FAIL(ctx, NGSID_ERR_ARG, "null argument");
    if (backbones->n != n_groups) NGSID_FAIL(ctx, NGSID_ERR_ARG, "one backbone per group expected");
    HostTimer ht(ctx->stream, "polish");
    DevReads RD; int32_t rc = ngsid_upload_reads(ctx, reads, &RD, false); if (rc) return rc;
    if (!read_order && grp_off[n_groups] > RD.n) NGSID_FAIL(ctx, NGSID_ERR_ARG, "group offsets exceed the read set");
    if (read_order) for (uint64_t x = 0; x < grp_off[n_groups]; ++x) if (read_order[x] >= RD.n) NGSID_FAIL(ctx, NGSID_ERR_ARG, "read_order[%llu] out of range", (unsigned long long)x);
    const uint64_t N = RD.n; const uint32_t G = (uint32_t)n_groups;
    const int W = prm->window > 0 ? prm->window : 500;
    ht.mark("upload + checks");
    // backbones to host strings (they are tiny and are rebuilt on the host after every iteration)
    std::vector<std::string> B(G);
    {
        std::vector<uint64_t> boff(G + 1); std::vector<uint8_t> bseq;
        if (backbones->mem == NGSID_MEM_DEVICE) {
            HIPCHK(ctx, hipMemcpy(boff.data(), backbones->off, 8 * (G + 1), hipMemcpyDeviceToHost)); bseq.resize(boff[G] + 1);
            if (boff[G]) HIPCHK(ctx, hipMemcpy(bseq.data(), backbones->seq, boff[G], hipMemcpyDeviceToHost));
        } else { memcpy(boff.data(), backbones->off, 8 * (G + 1)); bseq.assign(backbones->seq, backbones->seq + boff[G]); bseq.push_back(0); }
        for (uint32_t g = 0; g < G; ++g) B[g].assign((const char*)bseq.data() + boff[g], (size_t)(boff[g + 1] - boff[g]));
    }
    if (N == 0 || G == 0) {
        uint64_t total = 0; out_off[0] = 0; bool ovf = false;
        for (uint32_t g = 0; g < G; ++g) { if (trace) {} else if (total + B[g].size() <= out_cap && out) memcpy(out + total, B[g].data(), B[g].size()); else if (B[g].size()) ovf = true; total += B[g].size(); out_off[g + 1] = total; if (n_used) n_used[g] = 0; }
        if (trace) for (int it = 0; it < prm->iters; ++it) for (uint32_t g = 0; g < G; ++g) { trace->seq.push_back(B[g]); trace->used.push_back(0); }
        if (trace && trace->aln) for (uint64_t x = 0; x < (uint64_t)prm->iters * grp_off[n_groups] * 6; ++x) trace->aln[x] = -1;
        if (needed) *needed = total; if (ovf) NGSID_FAIL(ctx, NGSID_ERR_CAPACITY, "output buffer too small"); return NGSID_OK;
    }
    // ---- read -> group map, mean read length per group (TGS/NGS window type)
    std::vector<uint32_t> h_rgroup(N, 0xffffffffu); std::vector<uint8_t> tgs(G, 0);
    for (uint32_t g = 0; g < G; ++g) {
        double tot = 0; const uint64_t ns = grp_off[g + 1] - grp_off[g];
        for (uint64_t x = grp_off[g]; x < grp_off[g + 1]; ++x) {
            const uint64_t r = read_order ? read_order[x] : x;
            if (h_rgroup[r] != 0xffffffffu) NGSID_FAIL(ctx, NGSID_ERR_ARG, "read %llu is listed twice (groups %u and %u): strand and layers are kept per read, list every read under one backbone", (unsigned long long)r, h_rgroup[r], g);
            h_rgroup[r] = g; tot += (double)(RD.h_off[r + 1] - RD.h_off[r]);
        }
        tgs[g] = ns > 0 && (tot / (double)ns) > 1000.0;
    }
    ht.mark("group map");
    // ---- strand detection (replaces minimap2's strand call): shared HPC minimizers with the initial backbone, fw vs rc
    DevBuf<uint64_t>& mzcode = ctx->pol_mzcode; DevBuf<uint32_t>& mzcnt = ctx->mzc_cnt; DevBuf<uint32_t>& hlen = ctx->mzc_hlen; DevBuf<uint32_t> d_rgroup; DevBuf<double> herr, rawerr; DevBuf<uint8_t> d_orient; DevBuf<int> flag;
    HIPCHK(ctx, mzcnt.reserve(N)); HIPCHK(ctx, hlen.reserve(N)); HIPCHK(ctx, flag.alloc(1));
    HIPCHK(ctx, d_rgroup.alloc(N)); HIPCHK(ctx, d_orient.alloc(N));
    HIPCHK(ctx, hipMemcpyAsync(d_rgroup.p, h_rgroup.data(), 4 * N, hipMemcpyHostToDevice, ctx->stream));
    const int sk = std::min(prm->k, 21), sw = std::max(prm->w, sk);       // strand detection only needs SOME minimizer scheme: one-word codes, comparable between the two launches
    {   // the clustering call that preceded this one left the minimizers of the same reads in the context (same bases, offsets, k, w): reuse them
        bool hit = false;
        if (ctx->mzc.valid && ctx->mzc.n == N && ctx->mzc.total == RD.total && ctx->mzc.k == sk && ctx->mzc.w == sw) {
            unsigned long long fp = 0; rc = ngsid_reads_fingerprint(ctx, RD, &fp); if (rc) return rc;
            hit = fp == ctx->mzc.fp;
        }
        if (!hit) {
            ctx->mzc.valid = false;
            HIPCHK(ctx, herr.alloc(N)); HIPCHK(ctx, rawerr.alloc(N));
            static thread_local PinVec<uint32_t> h_c, h_l; h_c.resize(N); h_l.resize(N); long long bad = -1;
            rc = ngsid_minimizers_csr(ctx, RD, sk, sw, ngsid_ctx_mz(ctx), mzcnt.p, hlen.p, herr.p, rawerr.p, h_c.data(), h_l.data(), &bad); if (rc) return rc;
            if (bad >= 0) NGSID_FAIL(ctx, NGSID_ERR_ALPHABET, "base outside ACGTN in a read (read %lld)", bad);       // (the cache stays invalid: ADVICE r4)
            if (N >= 1024) { unsigned long long fp = 0; rc = ngsid_reads_fingerprint(ctx, RD, &fp); if (rc) return rc; ctx->mzc.n = N; ctx->mzc.total = RD.total; ctx->mzc.k = sk; ctx->mzc.w = sw; ctx->mzc.fp = fp; ctx->mzc.valid = true; }
        }
    }
    {
        // backbone fw + rc minimizers through the same kernel (a CSR of their own), then sorted on the host (a handful of short lists)
        std::vector<std::string> two; for (uint32_t g = 0; g < G; ++g) two.push_back(B[g]); for (uint32_t g = 0; g < G; ++g) two.push_back(revcomp(B[g]));
        std::vector<uint64_t> toff(2 * G + 1, 0); std::string cat; for (size_t i = 0; i < two.size(); ++i) { cat += two[i]; toff[i + 1] = cat.size(); }
        ngsid_reads_t br{(const uint8_t*)cat.data(), nullptr, toff.data(), 2ull * G, NGSID_MEM_HOST, 0};
        DevReads BR; rc = ngsid_upload_reads(ctx, &br, &BR, false); if (rc) return rc;
        DevBuf<uint64_t> bc, boff_; DevBuf<uint32_t> bp_, bcnt, bhl; DevBuf<double> be, bw; static thread_local PinVec<uint64_t> hmo; static thread_local PinVec<uint32_t> hcnt, hhl; hcnt.resize(2 * G); hhl.resize(2 * G);
        HIPCHK(ctx, bcnt.alloc(2 * G)); HIPCHK(ctx, bhl.alloc(2 * G)); HIPCHK(ctx, be.alloc(2 * G)); HIPCHK(ctx, bw.alloc(2 * G));
        long long bad = -1;
        rc = ngsid_minimizers_csr(ctx, BR, sk, sw, MzOut{&bc, &bp_, &boff_, &hmo}, bcnt.p, bhl.p, be.p, bw.p, hcnt.data(), hhl.data(), &bad); if (rc) return rc;
        if (bad >= 0) NGSID_FAIL(ctx, NGSID_ERR_ALPHABET, "base outside ACGTN in a backbone");
        const uint64_t btot = hmo[2 * G];
        std::vector<uint64_t> hc(btot + 1);
        if (btot) HIPCHK(ctx, hipMemcpyAsync(hc.data(), bc.p, 8 * btot, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<uint64_t> lists, loff(2 * G + 1, 0);
        for (uint32_t i = 0; i < 2 * G; ++i) {
            const uint32_t c = hhl[i] >= (uint32_t)sk ? hcnt[i] : 0;
            std::vector<uint64_t> v(hc.begin() + hmo[i], hc.begin() + hmo[i] + c); std::sort(v.begin(), v.end());
            lists.insert(lists.end(), v.begin(), v.end()); loff[i + 1] = lists.size();
        }
        DevBuf<uint64_t> d_lists, d_loff; HIPCHK(ctx, d_lists.alloc(lists.size() + 1)); HIPCHK(ctx, d_loff.alloc(loff.size()));
        if (!lists.empty()) HIPCHK(ctx, hipMemcpyAsync(d_lists.p, lists.data(), 8 * lists.size(), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(d_loff.p, loff.data(), 8 * loff.size(), hipMemcpyHostToDevice, ctx->stream));
        { ProfScope ps_(ctx, "k_strand"); hipLaunchKernelGGL(k_strand, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, ctx->stream, ctx->mz_off.p, mzcnt.p, hlen.p, mzcode.p, sk, d_rgroup.p, d_lists.p, d_loff.p, G, N, d_orient.p); }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    static thread_local PinVec<uint8_t> h_orient; h_orient.resize(N);
    HIPCHK(ctx, hipMemcpyAsync(h_orient.data(), d_orient.p, N, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ht.mark("minimizers + strand");
    // ---- oriented copies of the reads
    DevBuf<uint8_t>& oseq = ctx->pol_oseq; DevBuf<uint8_t>& oqual = ctx->pol_oqual; HIPCHK(ctx, oseq.reserve(RD.total + 16)); if (RD.qual) HIPCHK(ctx, oqual.reserve(RD.total + 16));
    { ProfScope ps_(ctx, "k_orient"); hipLaunchKernelGGL(k_orient, dim3((unsigned)N), dim3(128), 0, ctx->stream, RD.seq, RD.qual, RD.off, N, d_orient.p, oseq.p, RD.qual ? oqual.p : nullptr); }
    HIPCHK(ctx, hipGetLastError());
    // ---- pairs (usable reads of reads that belong to a group), fixed over the iterations
    static thread_local PinVec<uint32_t> pair_read, pair_group; pair_read.clear(); pair_group.clear();
    const bool want_aln = trace && trace->aln; const uint64_t NL = grp_off[n_groups];
    std::vector<uint64_t> pair_pos;                    // (alignment trace) position of a pair's read in the caller's list
    for (uint64_t x = 0; x < grp_off[n_groups]; ++x) { const uint64_t r = read_order ? read_order[x] : x; if (h_orient[r] != 255) { pair_read.push_back((uint32_t)r); pair_group.push_back(h_rgroup[r]); if (want_aln) pair_pos.push_back(x); } }
    DevBuf<int32_t> d_dist; static thread_local PinVec<int32_t> h_span, h_dist;
    if (want_aln) { HIPCHK(ctx, d_dist.alloc(pair_read.size() + 1)); h_span.resize(pair_read.size() * 4 + 4); h_dist.resize(pair_read.size() + 1); }
    uint64_t NP = pair_read.size();
    DevBuf<uint32_t> d_pair_read, d_pair_group; DevBuf<int32_t> d_open, d_span, d_blen; DevBuf<int32_t>& d_bp = ctx->pol_bp; DevBuf<uint8_t>& d_lay_raw = ctx->pol_lay; DevBuf<uint16_t>& d_valid = ctx->pol_valid;
    HIPCHK(ctx, d_pair_read.alloc(NP)); HIPCHK(ctx, d_pair_group.alloc(NP)); HIPCHK(ctx, d_open.alloc(NP)); HIPCHK(ctx, d_span.alloc(NP * 4)); HIPCHK(ctx, d_blen.alloc(G));
    if (NP) { HIPCHK(ctx, hipMemcpyAsync(d_pair_read.p, pair_read.data(), 4 * NP, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(ctx, hipMemcpyAsync(d_pair_group.p, pair_group.data(), 4 * NP, hipMemcpyHostToDevice, ctx->stream));
              HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)d_open.p, prm->aln_open, NP, ctx->stream)); }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> used(G, 0);
    ht.mark("orient + pairs");

    std::vector<uint8_t> stable(G, 0);              // stop_when_stable: groups whose last iteration returned the backbone unchanged
    for (int it = 0; it < prm->iters; ++it) {
        if (prm->stop_when_stable && it > 0) {
            bool any_active = false; for (uint32_t g = 0; g < G; ++g) any_active = any_active || !stable[g];
            if (!any_active) break;
            // drop the pairs of the stable groups (their reads would be aligned and stacked into exactly the same windows again)
            size_t keep = 0;
            for (size_t p = 0; p < pair_read.size(); ++p) if (!stable[pair_group[p]]) { pair_read[keep] = pair_read[p]; pair_group[keep] = pair_group[p]; if (want_aln) pair_pos[keep] = pair_pos[p]; ++keep; }
            if (keep != pair_read.size()) {
                pair_read.resize(keep); pair_group.resize(keep); NP = keep; if (want_aln) pair_pos.resize(keep);
                if (NP) { HIPCHK(ctx, hipMemcpyAsync(d_pair_read.p, pair_read.data(), 4 * NP, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(ctx, hipMemcpyAsync(d_pair_group.p, pair_group.data(), 4 * NP, hipMemcpyHostToDevice, ctx->stream)); HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); }
            }
        }
        // upload the current backbones
        std::vector<uint64_t> boff(G + 1, 0); std::string cat; std::vector<int32_t> blen(G); int nwinmax = 1; uint32_t maxb = 0;
        for (uint32_t g = 0; g < G; ++g) { cat += B[g]; boff[g + 1] = cat.size(); blen[g] = (int32_t)B[g].size(); nwinmax = std::max(nwinmax, (int)((B[g].size() + W - 1) / W)); maxb = std::max<uint32_t>(maxb, (uint32_t)B[g].size()); }
        ngsid_reads_t br{(const uint8_t*)cat.data(), nullptr, boff.data(), G, NGSID_MEM_HOST, 0};
        DevReads BB; rc = ngsid_upload_reads(ctx, &br, &BB, false); if (rc) return rc;
        HIPCHK(ctx, hipMemcpyAsync(d_blen.p, blen.data(), 4 * G, hipMemcpyHostToDevice, ctx->stream));
        for (uint32_t g = 0; g < G; ++g) if (!stable[g]) used[g] = 0;
        std::vector<Unit> units; std::vector<PSeq> bbs; std::vector<int> bb_len; std::vector<std::pair<uint32_t, int>> unit_gw;
        static thread_local PinVec<uint16_t> h_valid; int max_layer = 1;
        HIPCHK(ctx, hipMemsetAsync(flag.p, 0, sizeof(int), ctx->stream));
        if (NP) {
            HIPCHK(ctx, d_bp.reserve(NP * (uint64_t)nwinmax * 4)); HIPCHK(ctx, d_lay_raw.reserve(sizeof(PSeq) * NP * (uint64_t)nwinmax)); HIPCHK(ctx, d_valid.reserve(NP * (uint64_t)nwinmax));
            AlignJob J{};
            J.qseq = oseq.p; J.qoff = RD.off; J.tseq = BB.seq; J.toff = BB.off; J.qidx = d_pair_read.p; J.tidx = d_pair_group.p; J.npairs = NP;
            J.match = prm->aln_match; J.mismatch = prm->aln_mismatch; J.ext = prm->aln_ext; J.k = 1; J.open = d_open.p; J.match_id = nullptr;
            J.score = nullptr; J.ncols = nullptr; J.nmatch = nullptr; J.region = nullptr; J.bp = d_bp.p; J.bp_windows = nwinmax; J.window = W; J.span = d_span.p;
            int aln_mode = prm->aln_mode == 2 ? 1 : prm->aln_mode;
            if (aln_mode == 3) { aln_mode = 1; J.clip = 1; }          // edit distance + overlap-span clipping (include/ngsid.h)
            if (aln_mode == 1) rc = ngsid_launch_ed_align(ctx, J, RD.maxlen, maxb, want_aln ? d_dist.p : nullptr);          // unit-cost, bit-parallel (k_ed_align.hip)
            else rc = ngsid_launch_align(ctx, J, RD.maxlen, maxb, prm->aln_open);
            if (rc) return rc;
            const uint64_t T = NP * (uint64_t)nwinmax;
            { ProfScope ps_(ctx, "k_layers"); hipLaunchKernelGGL(k_layers, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, ctx->stream, oseq.p, RD.qual ? oqual.p : nullptr, RD.off, d_pair_read.p, d_pair_group.p, NP, nwinmax,
                               d_bp.p, d_span.p, d_blen.p, W, prm->quality_threshold, prm->error_threshold, (PSeq*)d_lay_raw.p, d_valid.p, flag.p); }
            HIPCHK(ctx, hipGetLastError());
            h_valid.resize(T);
            HIPCHK(ctx, hipMemcpyAsync(&max_layer, flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipMemcpyAsync(h_valid.data(), d_valid.p, 2 * T, hipMemcpyDeviceToHost, ctx->stream));
            if (want_aln) { HIPCHK(ctx, hipMemcpyAsync(h_span.data(), d_span.p, 16 * NP, hipMemcpyDeviceToHost, ctx->stream)); if (aln_mode == 1) HIPCHK(ctx, hipMemcpyAsync(h_dist.data(), d_dist.p, 4 * NP, hipMemcpyDeviceToHost, ctx->stream)); }
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (want_aln) {        // this iteration's records: the previous iteration's for the groups that are stable (same backbone, same reads: the same alignments), fresh ones for the pairs just aligned
                int32_t* A = trace->aln + (size_t)it * NL * 6;
                if (it == 0) for (uint64_t x = 0; x < NL * 6; ++x) A[x] = -1; else memcpy(A, A - NL * 6, sizeof(int32_t) * NL * 6);
                for (uint64_t p = 0; p < NP; ++p) {
                    int32_t* a = A + pair_pos[p] * 6; const uint32_t r = pair_read[p]; const int32_t ql = (int32_t)(RD.h_off[r + 1] - RD.h_off[r]);
                    const int32_t qf = h_span[p * 4], qe = h_span[p * 4 + 1], tf = h_span[p * 4 + 2], te = h_span[p * 4 + 3];
                    if (qf < 0) { for (int c = 0; c < 6; ++c) a[c] = -1; continue; }
                    const bool rcs = h_orient[r] == 1;
                    a[0] = rcs ? 1 : 0; a[1] = rcs ? ql - 1 - qe : qf; a[2] = rcs ? ql - qf : qe + 1; a[3] = tf; a[4] = te + 1; a[5] = aln_mode == 1 ? h_dist[p] : -1;
                }
            }
        } else if (want_aln) { int32_t* A = trace->aln + (size_t)it * NL * 6; if (it == 0) for (uint64_t x = 0; x < NL * 6; ++x) A[x] = -1; else memcpy(A, A - NL * 6, sizeof(int32_t) * NL * 6); }
        ht.mark("align + layers + valid copy");
        // ---- units = (group, window) with their layers in read order
        std::vector<std::vector<int>> unit_of(G);
        for (uint32_t g = 0; g < G; ++g) { if (stable[g]) continue; const int nw = polish_nwin((int)B[g].size(), W); unit_of[g].assign(nw, -1);
            for (int wdx = 0; wdx < nw; ++wdx) { unit_of[g][wdx] = (int)units.size(); units.emplace_back(); unit_gw.push_back({g, wdx}); } }
        {   // layers of every window in pair (= read) order; the units of a group are consecutive, a window holds at most one layer per pair of its group
            std::vector<uint32_t> ubase(G), unw(G), npg(G, 0);
            for (uint32_t g = 0; g < G; ++g) { unw[g] = (uint32_t)unit_of[g].size(); ubase[g] = unw[g] ? (uint32_t)unit_of[g][0] : 0; }
            for (uint64_t p = 0; p < NP; ++p) npg[pair_group[p]]++;
            const uint32_t* pg = pair_group.data(); const uint16_t* hv0 = h_valid.data();
            // pairs come group by group (the caller's read lists); then every (group, window) unit is filled and ordered by its own host thread
            // from the group's pair range.  Otherwise (never seen) one pass over all pairs.
            bool by_group = true; std::vector<uint64_t> gbeg(G + 1, 0);
            for (uint64_t p = 1; p < NP; ++p) if (pg[p] < pg[p - 1]) { by_group = false; break; }
            if (by_group) { uint64_t acc = 0; for (uint32_t g = 0; g < G; ++g) { gbeg[g] = acc; acc += npg[g]; } gbeg[G] = acc; }
            // racon adds the layers of a window in the order of their first window position (src/window.cpp: rank sorted by positions_.first);
            // ties keep the read order (stable): counting sort on the window position
            auto order_unit = [&](std::vector<uint32_t>& v, std::vector<uint32_t>& tmpv, std::vector<uint32_t>& cnt) {
                if (v.size() < 2) return;
                uint32_t kmax = 0; bool sorted = true;
                for (size_t x = 0; x < v.size(); ++x) { const uint32_t kx = hv0[v[x]]; kmax = std::max(kmax, kx); if (x && hv0[v[x - 1]] > kx) sorted = false; }
                if (sorted) return;
                cnt.assign((size_t)kmax + 2, 0);
                for (uint32_t id : v) cnt[hv0[id] + 1]++;
                for (size_t k2 = 1; k2 < cnt.size(); ++k2) cnt[k2] += cnt[k2 - 1];
                tmpv.resize(v.size());
                for (uint32_t id : v) tmpv[cnt[hv0[id]]++] = id;
                std::copy(tmpv.begin(), tmpv.end(), v.begin());
            };
            if (by_group && NP >= 65536) {
                std::vector<uint64_t> used_g(G, 0);
                const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
                const size_t ntask = units.size() + G;                          // one task per unit + one per group (reads used)
                std::atomic<size_t> next_task{0};
                auto worker = [&]() {
                    std::vector<uint32_t> tmpv, cnt;
                    for (;;) {
                        const size_t t = next_task.fetch_add(1); if (t >= ntask) break;
                        if (t < units.size()) {
                            const uint32_t g = unit_gw[t].first, wdx = (uint32_t)unit_gw[t].second;
                            auto& v = units[t].seqs; v.resize(npg[g]); uint32_t* c = v.data();
                            for (uint64_t p = gbeg[g]; p < gbeg[g + 1]; ++p) { *c = (uint32_t)(p * (uint64_t)nwinmax + wdx); c += hv0[p * (uint64_t)nwinmax + wdx] != 0; }
                            v.resize((size_t)(c - v.data()));
                            order_unit(v, tmpv, cnt);
                        } else {
                            const uint32_t g = (uint32_t)(t - units.size()); const uint32_t nwg = unw[g]; uint64_t u = 0;
                            for (uint64_t p = gbeg[g]; p < gbeg[g + 1]; ++p) { const uint16_t* hv = hv0 + p * (uint64_t)nwinmax; uint32_t any = 0; for (uint32_t wdx = 0; wdx < nwg; ++wdx) any |= hv[wdx] != 0; u += any; }
                            used_g[g] = u;
                        }
                    }
                };
                std::vector<std::thread> th; for (unsigned x = 1; x < hw; ++x) th.emplace_back(worker);
                worker(); for (auto& t : th) t.join();
                for (uint32_t g = 0; g < G; ++g) used[g] += (uint32_t)used_g[g];
            } else {
                // lists sized for the worst case, filled through raw cursors, trimmed afterwards
                std::vector<uint32_t*> cur(units.size(), nullptr);
                for (uint32_t g = 0; g < G; ++g) for (uint32_t wdx = 0; wdx < unw[g]; ++wdx) { auto& v = units[ubase[g] + wdx].seqs; v.resize(npg[g]); cur[ubase[g] + wdx] = v.data(); }
                for (uint64_t p = 0; p < NP; ++p) {
                    const uint32_t g = pg[p]; const uint16_t* hv = hv0 + p * (uint64_t)nwinmax; const uint32_t nwg = unw[g], ub = ubase[g]; uint32_t any = 0;
                    for (uint32_t wdx = 0; wdx < nwg; ++wdx) { const uint32_t v = hv[wdx] != 0; *cur[ub + wdx] = (uint32_t)(p * (uint64_t)nwinmax + wdx); cur[ub + wdx] += v; any |= v; }
                    used[g] += any;
                }
                for (size_t u = 0; u < units.size(); ++u) if (cur[u]) units[u].seqs.resize((size_t)(cur[u] - units[u].seqs.data()));
                std::vector<uint32_t> tmpv, cnt;
                for (size_t u = 0; u < units.size(); ++u) order_unit(units[u].seqs, tmpv, cnt);
            }
        }
        std::vector<size_t> nlayers(units.size());
        for (size_t u = 0; u < units.size(); ++u) {
            nlayers[u] = units[u].seqs.size();
            if (units[u].seqs.size() < 2) { units[u].seqs.clear(); units[u].done = true; continue; }          // racon: < 3 sequences incl. backbone -> keep backbone
            const uint32_t g = unit_gw[u].first; const int ws = unit_gw[u].second * W; const int wlen = polish_wlen((int)B[g].size(), W, unit_gw[u].second);
            PSeq S; S.s = BB.seq + boff[g] + ws; S.q = nullptr; S.len = wlen; S.uw = 0; S.cw = 0; S.mode = NGSID_POA_GLOBAL; S.a0 = 0; S.a1 = -1;
            units[u].bb = (int)bbs.size(); bbs.push_back(S); bb_len.push_back(wlen);
        }
        DevBuf<PSeq> d_bbs; HIPCHK(ctx, d_bbs.alloc(bbs.size()));
        if (!bbs.empty()) HIPCHK(ctx, hipMemcpyAsync(d_bbs.p, bbs.data(), sizeof(PSeq) * bbs.size(), hipMemcpyHostToDevice, ctx->stream));
        bool any_tgs = prm->trim == 2; for (uint32_t g = 0; g < G; ++g) any_tgs = any_tgs || (tgs[g] && prm->trim);
        HierParams hp{prm->match, prm->mismatch, prm->gap, prm->band > 0 ? prm->band : (RD.maxlen <= NGSID_POA_BAND64_MAXLEN ? 64 : 128), prm->node_cap, prm->tile_depth, NGSID_POA_GLOBAL, any_tgs, (prm->trim >= 2 ? 1 : 0) | (prm->trim == 3 ? 4 : 0)};      // trim_tiles: 1 = trim tile consensuses, 4 = except the tile that ends a unit (trim 3)
        hp.single_below = prm->single_below > 0 ? prm->single_below : 0;
        if (hp.single_below > 0) for (size_t u = 0; u < units.size(); ++u) {       // windows with few layers: ONE graph; the longest sequence that can enter it = the longest READ behind its layers, or the window
            Unit& U = units[u]; if (U.done || U.seqs.size() >= (size_t)hp.single_below) continue;
            uint32_t mx = (uint32_t)std::max(1, bb_len[U.bb]);
            for (uint32_t id : U.seqs) { const uint32_t r = pair_read[id / (uint32_t)nwinmax]; mx = std::max<uint32_t>(mx, (uint32_t)(RD.h_off[r + 1] - RD.h_off[r])); }
            U.single_maxlen = mx;
        }
        ht.mark("unit lists");
        rc = run_hierarchy(ctx, (const PSeq*)d_lay_raw.p, (uint32_t)std::max(max_layer, 1), d_bbs.p, bb_len, units, hp); if (rc) return rc;
        ht.mark("hierarchy");
        // ---- new backbones
        std::vector<std::string> NB(G);
        for (size_t u = 0; u < units.size(); ++u) {
            const uint32_t g = unit_gw[u].first; const int ws = unit_gw[u].second * W; const int wlen = polish_wlen((int)B[g].size(), W, unit_gw[u].second);
            std::string c = units[u].has_result ? units[u].result : std::string();
            if (!c.empty() && prm->trim && (tgs[g] || prm->trim == 2) && units[u].cov.size() == c.size()) {      // trim 3: racon's window rule (TGS only) with trimmed tiles
                const uint32_t avg = (uint32_t)(nlayers[u] / 2); int b = 0, e = (int)c.size() - 1;
                for (; b < (int)c.size(); ++b) if (units[u].cov[b] >= avg) break;
                for (; e >= 0; --e) if (units[u].cov[e] >= avg) break;
                if (b < e) c = c.substr(b, e - b + 1);
            }
            if (c.empty()) c = B[g].substr(ws, wlen);
            NB[g] += c;
        }
        for (uint32_t g = 0; g < G; ++g) { if (stable[g]) NB[g] = B[g]; else if (prm->stop_when_stable && NB[g] == B[g]) stable[g] = 1; }
        B.swap(NB);
        if (trace) for (uint32_t g = 0; g < G; ++g) { trace->seq.push_back(B[g]); trace->used.push_back(used[g]); }
    }
    if (trace && trace->aln) for (size_t it = trace->seq.size() / std::max<uint32_t>(G, 1); it < (size_t)prm->iters; ++it) { if (it == 0) { for (uint64_t x = 0; x < grp_off[n_groups] * 6; ++x) trace->aln[x] = -1; } else memcpy(trace->aln + it * grp_off[n_groups] * 6, trace->aln + (it - 1) * grp_off[n_groups] * 6, sizeof(int32_t) * grp_off[n_groups] * 6); }
    if (trace) while (trace->seq.size() < (size_t)prm->iters * G) { const size_t x = trace->seq.size() - G; trace->seq.push_back(trace->seq[x]); trace->used.push_back(trace->used[x]); }      // every group stable: the remaining iterations return the same strings
    uint64_t total = 0; bool ovf = false; out_off[0] = 0;
    for (uint32_t g = 0; g < G; ++g) {
        if (trace) { /* ngsid_polish_trace: the final sequences are the trace's last iteration, no buffer was handed in */ }
        else if (total + B[g].size() <= out_cap && out) memcpy(out + total, B[g].data(), B[g].size()); else if (B[g].size()) ovf = true;
        total += B[g].size(); out_off[g + 1] = total; if (n_used) n_used[g] = used[g];
    }
    if (needed) *needed = total;
    if (ovf) NGSID_FAIL(ctx, NGSID_ERR_CAPACITY, "output buffer too small: need %llu bytes", (unsigned long long)total);
    return NGSID_OK;
}
