/* ngsid_merge_schedule.h - the merge rounds of the reference's parallel_clustering (modules/parallelize.py:169-217) on gathered
 * representatives, as plain C over a clustering callback.  Shared by libngsid_hip.so (ngsid_merge_representatives: callback =
 * ngsid_cluster_greedy on the GPU) and by the test oracle (callback = its scalar restatement); it is schedule bookkeeping only - which
 * representatives meet in which round, in which order, and who keeps its database - and holds no arithmetic of the path.
 *
 * State on entry = after round 1 of `--t N` / after every GPU clustered its own shard: R surviving representatives, each with its batch
 * index (1-based), score and HPC error rate.  Rounds: representatives sorted by (batch, score descending) are grouped pairwise by batch
 * (batch_list(..., merge_consecutive=True), parallelize.py:34-45: batches (1,2)(3,4)...); inside a group the reads of the LOWEST batch
 * index seed the database and are not re-clustered (cluster.py:221-223,243-248), the others are clustered against them in score order;
 * survivors get the group's new index; repeat until one batch is left, which is clustered once more (parallelize.py:141-147).
 */
#ifndef NGSID_MERGE_SCHEDULE_H
#define NGSID_MERGE_SCHEDULE_H
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "ngsid.h"

/* clusters the reads of `sub` (processing order); same contract as ngsid_cluster_greedy minus the ctx */
typedef int32_t (*ngsid_cluster_cb)(void* user, const ngsid_reads_t* sub, const ngsid_cluster_params_t* prm, const uint32_t* acc_rank,
                                    const int32_t* prev_batch, const double* known_err, int32_t* rep_of_read, double* hpc_err_out, uint8_t* status_out);

typedef struct { int64_t batch; double score; int64_t idx; } ngsid_ms_key;
static int ngsid_ms_cmp(const void* a, const void* b)
{
    const ngsid_ms_key* x = (const ngsid_ms_key*)a; const ngsid_ms_key* y = (const ngsid_ms_key*)b;
    if (x->batch != y->batch) return x->batch < y->batch ? -1 : 1;
    if (x->score != y->score) return x->score > y->score ? -1 : 1;       /* score descending */
    return x->idx < y->idx ? -1 : (x->idx > y->idx);                     /* stable: original order */
}

/* reps: HOST CSR with qualities.  batch[R] (1-based), score[R], hpc_err[R] (NaN = unknown), acc_rank[R].  rep_of[R] out: index of the final
 * representative of every gathered representative (itself if it survives). */
static int32_t ngsid_merge_schedule(ngsid_cluster_cb cb, void* user, const ngsid_reads_t* reps, const ngsid_cluster_params_t* prm,
                                    const uint32_t* acc_rank, const double* score, const double* hpc_err, const int32_t* batch, int32_t n_batches,
                                    int32_t* rep_of)
{
    const int64_t R = (int64_t)reps->n;
    for (int64_t i = 0; i < R; ++i) rep_of[i] = (int32_t)i;
    if (R == 0 || n_batches <= 1) return NGSID_OK;
    int64_t* bidx = (int64_t*)malloc(sizeof(int64_t) * (size_t)R); double* herr = (double*)malloc(sizeof(double) * (size_t)R);
    uint8_t* alive = (uint8_t*)malloc((size_t)R); ngsid_ms_key* keys = (ngsid_ms_key*)malloc(sizeof(ngsid_ms_key) * (size_t)R);
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)R);
    for (int64_t i = 0; i < R; ++i) { bidx[i] = batch[i]; herr[i] = hpc_err ? hpc_err[i] : NAN; alive[i] = 1; }
    uint64_t total = reps->off[R] - reps->off[0];
    uint8_t* sseq = (uint8_t*)malloc((size_t)total + 1); uint8_t* squal = (uint8_t*)malloc((size_t)total + 1); uint64_t* soff = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)R + 1));
    uint32_t* srank = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)R); int32_t* sprev = (int32_t*)malloc(sizeof(int32_t) * (size_t)R); double* sknown = (double*)malloc(sizeof(double) * (size_t)R);
    int32_t* srep = (int32_t*)malloc(sizeof(int32_t) * (size_t)R); double* sherr = (double*)malloc(sizeof(double) * (size_t)R); uint8_t* sst = (uint8_t*)malloc((size_t)R);
    int32_t rc = NGSID_OK;
    for (;;) {
        /* the alive representatives, by (batch, score descending), grouped pairwise by batch index */
        int64_t n = 0;
        for (int64_t i = 0; i < R; ++i) if (alive[i]) { keys[n].batch = bidx[i]; keys[n].score = score[i]; keys[n].idx = i; ++n; }
        qsort(keys, (size_t)n, sizeof(ngsid_ms_key), ngsid_ms_cmp);
        for (int64_t x = 0; x < n; ++x) order[x] = keys[x].idx;
        /* batch_list(..., merge_consecutive=True) (parallelize.py:34-45), quirks included: an element whose batch index exceeds batch_id closes the
           current group (even an empty one), raises batch_id by 2 ONCE and opens the next group with itself; empty groups keep their number */
        int64_t* gstart = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n + 2)); int64_t ngroups = 0;
        { int64_t bid = 2; gstart[ngroups++] = 0; for (int64_t x = 0; x < n; ++x) if (bidx[order[x]] > bid) { gstart[ngroups++] = x; bid += 2; } gstart[ngroups] = n; }
        const int single = ngroups == 1;
        for (int64_t gi = 0; gi < ngroups && rc == NGSID_OK; ++gi) {
            const int64_t a = gstart[gi], b = gstart[gi + 1];
            if (b > a) {
                const int64_t m = b - a; const int64_t new_index = single ? 1 : gi + 1;
                uint64_t o = 0;
                for (int64_t x = 0; x < m; ++x) {
                    const int64_t i = order[a + x]; const uint64_t l = reps->off[i + 1] - reps->off[i];
                    memcpy(sseq + o, reps->seq + reps->off[i], (size_t)l); memcpy(squal + o, reps->qual + reps->off[i], (size_t)l);
                    soff[x] = o; o += l; srank[x] = acc_rank ? acc_rank[i] : (uint32_t)i; sprev[x] = (int32_t)bidx[i]; sknown[x] = herr[i];
                }
                soff[m] = o;
                ngsid_reads_t sub; sub.seq = sseq; sub.qual = squal; sub.off = soff; sub.n = (uint64_t)m; sub.mem = NGSID_MEM_HOST; sub._pad = 0;
                rc = cb(user, &sub, prm, srank, sprev, sknown, srep, sherr, sst);
                if (rc != NGSID_OK) break;
                for (int64_t x = 0; x < m; ++x) {
                    const int64_t i = order[a + x];
                    if (srep[x] != (int32_t)x) { rep_of[i] = (int32_t)order[a + srep[x]]; alive[i] = 0; }
                    else { if (!isnan(sherr[x])) herr[i] = sherr[x]; if (sst[x] != NGSID_ST_SHORT) bidx[i] = new_index; }
                }
            }
        }
        free(gstart);
        if (rc != NGSID_OK || single) break;
    }
    if (rc == NGSID_OK) for (int pass = 0; pass < 64; ++pass) {      /* a representative that joined later drags its cluster (cluster.py:338-345) */
        int changed = 0;
        for (int64_t i = 0; i < R; ++i) { const int32_t r2 = rep_of[rep_of[i]]; if (r2 != rep_of[i]) { rep_of[i] = r2; changed = 1; } }
        if (!changed) break;
    }
    free(bidx); free(herr); free(alive); free(keys); free(order); free(sseq); free(squal); free(soff); free(srank); free(sprev); free(sknown); free(srep); free(sherr); free(sst);
    return rc;
}
#endif
