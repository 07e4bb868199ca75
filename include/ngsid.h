/* ngsid.h - C-ABI of libngsid_hip.so: the MI355X (gfx950) hot path of NGSpeciesID.
 *
 * This library replaces, for the read-clustering + consensus + polishing path only, what the
 * reference reaches through its four native third-party tools (parasail, spoa, minimap2, racon)
 * and the pure-Python loops around them.  Citations are file:line into ksahlin/NGSpeciesID v0.3.1.
 *
 * Conventions
 *   - plain C, POD only; every function returns int32 (0 = NGSID_OK, <0 = error code);
 *     ngsid_last_error(ctx) returns the text of the last failure on that ctx.
 *   - read sets are CSR: concatenated bytes + uint64 offsets[n+1]; `mem` says where the three
 *     pointers live (NGSID_MEM_HOST: the library copies them to HBM; NGSID_MEM_DEVICE: they already
 *     are HBM pointers on the ctx's device and are used in place).  Outputs are host pointers
 *     unless stated.  The library never frees caller memory and never keeps a caller pointer after
 *     the call returns.
 *   - one ctx = one HIP device + one HIP stream.  Not thread-safe across threads sharing a ctx.
 *   - there is NO CPU implementation behind these symbols: without a usable HIP device
 *     ngsid_create fails with NGSID_ERR_NO_DEVICE.  (The CPU restatement used for testing lives
 *     in oracle/ under the ongsid_* names and is never linked here.)
 */
#ifndef NGSID_H
#define NGSID_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NGSID_OK                 0
#define NGSID_ERR_NO_DEVICE     -1   /* no HIP device / kernel image not loadable */
#define NGSID_ERR_ARG           -2   /* bad argument (NULL, k>32, w<k, ...) */
#define NGSID_ERR_ALPHABET      -3   /* base outside {A,C,G,T,N} met by the minimizer encoder */
#define NGSID_ERR_CAPACITY      -4   /* caller buffer too small; required size reported in *needed */
#define NGSID_ERR_HIP           -5   /* HIP runtime error (text in ngsid_last_error) */
#define NGSID_ERR_TOO_LONG      -6   /* a read exceeds NGSID_MAX_READ_LEN (consensus stages: NGSID_MAX_CONSENSUS_LEN / the POA engine's bounds) */
#define NGSID_ERR_NO_PTABLE     -7   /* (e1,e2) looked up in a p_shared table with a NaN hole (KeyError in cluster.py:367) */

#define NGSID_MEM_HOST   0u
#define NGSID_MEM_DEVICE 1u

#define NGSID_MAX_K          32      /* k <= 21: 3-bit order-preserving k-mer codes (0=end,A,C,G,N,T) in a uint64; 22..32: two-word codes inside the
                                        library, handed on as dense order-preserving ranks per call (the reference's table has rows for k = 10..30) */
#define NGSID_MAX_READ_LEN   65535   /* bases per read: scoring, minimizers, clustering (round 5; reads above 16 384 bases run the minimizer kernel's LONG layout, pairs with a
                                        sequence above 4 000 bases the int32 aligner) */
#define NGSID_MAX_CONSENSUS_LEN 16384 /* bases per sequence in the edit-distance aligner / polisher; the POA engine's own bounds are tighter for local-mode scoring
                                        (match x length < 65 536: 13 107 bases at the reference's match = 5) - NGSID_ERR_TOO_LONG beyond */
#define NGSID_POA_BAND64_MAXLEN 3000 /* default POA band (ngsid_poa_params_t.band <= 0): 64 columns when every read of the call has at most this many bases, else 128 (round 4: was 1 024; measured on 2 kb reads at 0.1 / 5 / 10 % error the 64-column first attempt + the redo of the tiles that touch the band edge is 26 - 31 % faster, at 5 kb ONT it is a wash) */

typedef struct ngsid_ctx ngsid_ctx;

typedef struct {
    const uint8_t*  seq;    /* concatenated bases (ASCII) */
    const uint8_t*  qual;   /* concatenated phred+33 characters; may be NULL where a function says so */
    const uint64_t* off;    /* n+1 offsets into seq/qual */
    uint64_t        n;      /* number of reads */
    uint32_t        mem;    /* NGSID_MEM_HOST or NGSID_MEM_DEVICE */
    uint32_t        _pad;
} ngsid_reads_t;

/* Clustering parameters = the argparse flags the path reads (NGSpeciesID:225-233) + the selected
 * 15x15 slice of the empirical table (NGSpeciesID:72-77), p_shared[(i*15)+j] for e1=(i+1)/100,
 * e2=(j+1)/100, NaN where the reference dict would have no key. */
typedef struct {
    int32_t k, w, min_shared, symmetric;
    double  min_fraction, mapped_threshold, aligned_threshold, min_prob_no_hits;
    double  p_shared[225];
} ngsid_cluster_params_t;

/* status_out codes of ngsid_cluster_greedy */
#define NGSID_ST_NEWREP   0   /* read founded a cluster (cluster.py:328-334) */
#define NGSID_ST_MAPPED   1   /* joined by the mapping criterion (cluster.py:307-308) */
#define NGSID_ST_ALIGNED  2   /* joined by the block-alignment criterion (cluster.py:313-314) */
#define NGSID_ST_SHORT    3   /* HPC length < k: skipped, stays a singleton (cluster.py:266-268) */
#define NGSID_ST_SEEDED   4   /* prev_batch_index == lowest: pre-existing representative (cluster.py:243-248) */

int32_t     ngsid_create(int32_t device_ordinal, uint32_t flags, ngsid_ctx** out);
void        ngsid_destroy(ngsid_ctx* ctx);
const char* ngsid_last_error(ngsid_ctx* ctx);
/* ABI/version probe usable without a device (tests check the library loads and exports all symbols). */
uint32_t    ngsid_abi_version(void);

/* (f1) replaces get_sorted_fastq_for_cluster.calc_score_new / fastq_single_core :23-33,124-155.
 * score[i] = expected number of error-free k-mers, err_rate[i] = mean 10^-(q/10) (no clamp),
 * keep[i] = 0 when len<2k, HPC len<k or 10*-log10(err)<=q_threshold.  Limits: k <= 64 (NGSID_ERR_ARG), reads <= 65535 bases (NGSID_ERR_TOO_LONG). */
int32_t ngsid_score_reads(ngsid_ctx* ctx, const ngsid_reads_t* reads, int32_t k, double q_threshold,
                          double* score, double* err_rate, uint8_t* keep);

/* (a1-a3) replaces the inline HPC (cluster.py:265), get_kmer_minimizers (cluster.py:16-39) and the
 * HPC quality / error-rate block (cluster.py:279-291).  Output is CSR over reads in read order:
 * mz_off[n+1], codes/pos with capacity `cap` entries (NGSID_ERR_CAPACITY + *needed otherwise).
 * codes are 3-bit-per-base order-preserving k-mer codes for k <= 21; for 22 <= k <= 32 they are the dense ranks of the k-mers among all
 * minimizers of THIS call (order preserving, equal k-mers equal code; not comparable between calls); pos are positions in the HPC string.
 * hpc_len[i] = HPC length, hpc_err[i] = error rate (sum over quality characters in ascending
 * character code), reads with hpc_len<k get 0 minimizers.
 * codes/pos/mz_off follow reads->mem (device outputs for device inputs); hpc_len/hpc_err are host. */
int32_t ngsid_hpc_minimizers(ngsid_ctx* ctx, const ngsid_reads_t* reads, int32_t k, int32_t w,
                             uint64_t* mz_off, uint64_t* codes, uint32_t* pos, uint64_t cap, uint64_t* needed,
                             uint32_t* hpc_len, double* hpc_err);

/* (a4-a11) replaces cluster.reads_to_clusters (cluster.py:207-353) incl. get_all_hits :43-62,
 * get_best_cluster :67-127, get_best_cluster_block_align :172-205, parasail_block_alignment :130-169.
 * reads           score-ordered read set (the order IS the greedy order)
 * acc_rank[n]     rank of the accession string (incl. "_score" suffix) under byte-wise comparison,
 *                 equal strings equal rank: the third sort key of cluster.py:79,174
 * prev_batch[n]   previous batch index per read or NULL (= all 0, single_clustering NGSpeciesID:20-33);
 *                 reads whose index equals max(1,min(prev_batch)) seed the database and are not
 *                 re-clustered (cluster.py:221-223,243-248)
 * known_err[n]    HPC error rates carried over from an earlier round (8-tuples, cluster.py:273-277),
 *                 NaN / NULL = compute
 * rep_of_read[n]  out: index of the representative read each read ends up with (itself for
 *                 representatives, seeded and skipped reads) = cluster_to_new_cluster_id (:324)
 * hpc_err_out[n]  out: the error rate in the representative 8-tuple (NaN for NGSID_ST_SHORT)
 * status_out[n]   out: NGSID_ST_*
 * counters[4]     out: mapped_passed, aln_passed, aln_called (cluster.py:349-351), number of new representatives */
int32_t ngsid_cluster_greedy(ngsid_ctx* ctx, const ngsid_reads_t* reads, const ngsid_cluster_params_t* prm,
                             const uint32_t* acc_rank, const int32_t* prev_batch, const double* known_err,
                             int32_t* rep_of_read, double* hpc_err_out, uint8_t* status_out, uint64_t counters[4]);

/* (a10,a15) replaces parasail.sg_trace_scan_16/32 + cigar_to_seq + the k-window identity filter
 * (cluster.py:130-169) and the identity count of consensus.highest_aln_identity (consensus.py:129-145).
 * Pair p aligns query read q_idx[p] of `queries` (rows) against target t_idx[p] of `targets` (columns),
 * semi-global (all four end gaps free), substitution match/mismatch over ACGT (case-insensitive, any other
 * character scores 0), first gap base costs open[p], each further one ext.
 * Outputs per pair: score; n_cols = length of the two gapped strings (both sequences covered);
 * n_match = columns whose two characters are equal; region = number of k-column windows holding
 * >= match_id[p] equal columns (sum(aligned_region), cluster.py:147-167).  Any output may be NULL. */
int32_t ngsid_sg_align_batch(ngsid_ctx* ctx, const ngsid_reads_t* queries, const ngsid_reads_t* targets,
                             const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                             int32_t match, int32_t mismatch, const int32_t* open, int32_t ext,
                             int32_t k, const int32_t* match_id,
                             int32_t* score, int32_t* n_cols, int32_t* n_match, int32_t* region);

/* (a10, boundary 8b) the alignment itself - what parasail hands back as result.cigar (cluster.py:138-144, consensus.py:64-73): for pair p the
 * columns of the two gapped strings in alignment order, one byte each: '=' equal characters, 'X' different characters, 'I' query only, 'D' target
 * only; free end gaps are part of it (both sequences are covered), so ops_off[p+1] - ops_off[p] = n_cols of ngsid_sg_align_batch.  ops_off has
 * n_pairs+1 entries; NGSID_ERR_CAPACITY (+ *needed) when cap is too small (qlen + tlen per pair always suffices). */
int32_t ngsid_sg_align_cigar_batch(ngsid_ctx* ctx, const ngsid_reads_t* queries, const ngsid_reads_t* targets,
                                   const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                                   int32_t match, int32_t mismatch, const int32_t* open, int32_t ext,
                                   int32_t* score, uint64_t* ops_off, uint8_t* ops, uint64_t cap, uint64_t* needed);

/* (a17) the read->backbone alignment of the polisher in edit-distance mode (ngsid_polish_params_t.aln_mode = 1; racon obtains the
 * same thing from edlib, racon src/overlap.cpp find_breaking_points): pair p aligns all of query q_idx[p] inside target t_idx[p]
 * with unit costs (target ends free).  Rules as in ngsid_polish_params_t.aln_mode.  Outputs per pair (any may be NULL):
 * distance; span = {q_first, q_last, t_first, t_last} of the aligned (match/mismatch) columns, -1 if none; and, per polishing
 * window w of `window` target bases (w < bp_windows), bp[(p*bp_windows+w)*4..] = {q_first, q_last, t_first, t_last} of the aligned
 * columns whose target position falls in the window, -1 if none. */
int32_t ngsid_ed_align_batch(ngsid_ctx* ctx, const ngsid_reads_t* queries, const ngsid_reads_t* targets,
                             const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                             int32_t window, int32_t bp_windows, int32_t* distance, int32_t* span, int32_t* bp);

/* POA modes */
#define NGSID_POA_LOCAL   0   /* spoa -l 0 */
#define NGSID_POA_GLOBAL  1   /* spoa -l 1 (racon windows) */
#define NGSID_POA_SEMI    2   /* sequence end-to-end, graph ends free (racon sub-graph layers) */

typedef struct {
    int32_t mode;          /* NGSID_POA_* */
    int32_t match, mismatch, gap;   /* spoa -m/-n/-g ; linear gaps (g >= e in consensus.py:87) */
    int32_t tile_depth;    /* reads per exact-order POA tile; <=0 = one tile per group (exact spoa order) */
    int32_t band;          /* DP band width in columns (64/128/256) of the first attempt; <=0 = library default (64 when every read of the call has
                              <= NGSID_POA_BAND64_MAXLEN (3 000) bases, else 128).  A tile in which a traceback touches a clipped band edge is redone with twice the band (up to 256) */
    int32_t node_cap;      /* graph node capacity per tile as a multiple of 1/16 of the first read length (<=0 default) */
    int32_t trim;          /* 0 = none (spoa: the heaviest bundle is completed to a sink, so the consensus can carry the unsupported tail of a single
                              read); 1 = coverage-trim the ends of every tile consensus (bases covered by less than half of the sequences merged) AND,
                              on the upper levels of the hierarchy (members = weighted tile consensuses, the final tile included), drop interior bases
                              whose column carries less than a third of the merged weight (round 3; a genuine insertion carried by under a third of the
                              reads goes as well - with tile_depth <= 0 there is one level and the rule never applies).  DESIGN.md section 2 */
    int32_t single_below;  /* round 6: a group with FEWER sequences than this is aligned as ONE graph in file order - spoa's own order (consensus.py:257-266 hands a cluster's
                              reads to one spoa process) - whatever tile_depth says, with a graph capacity of NGSID_POA_SINGLE_NODE_CAP / 16 times its first sequence when the
                              longest sequence of the GROUP (polisher: the longest read behind a window's layers, or the window) has at most NGSID_POA_SINGLE_MAXLEN bases (else
                              node_cap): a rule of the group alone, so the result does not depend on what else the call holds.  Depth tiling is a throughput device for deep groups; below
                              ~100 sequences one graph is at least as accurate (profiles/r06_tile_depth_sweep.txt).  0 = off (every group is tiled at tile_depth) */
} ngsid_poa_params_t;
#define NGSID_POA_SINGLE_NODE_CAP 160      /* one-graph units: room for ten times the first sequence, so that no graph is closed early */
#define NGSID_POA_SINGLE_MAXLEN   4096     /* ... while 10 x 1.25 x the longest sequence stays inside the tile engine's 16-bit node indices */

/* (a13,a14) replaces form_draft_consensus' per-cluster `spoa reads.fq -l 0 -r 0 -g -2` (consensus.py:83-92,249-278).
 * Group g = positions [grp_off[g], grp_off[g+1]) of `read_order` (host array of read indices; NULL = identity),
 * in spoa file order (representative first), so clusters need no data movement.  qual==NULL means unit
 * weights (FASTA input).  Consensus strings come back CSR: cons_off[n_groups+1] + cons bytes (capacity cons_cap). */
int32_t ngsid_poa_consensus(ngsid_ctx* ctx, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                            const ngsid_poa_params_t* prm,
                            uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed);

/* same + per-base coverage: cov[x] = number of reads (count weight) with a base in the alignment column of consensus base x, CSR-aligned with cons */
int32_t ngsid_poa_consensus_cov(ngsid_ctx* ctx, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint32_t* cov, uint64_t cons_cap, uint64_t* needed);

/* (8e step 3: the N-GPU merge of per-shard partial consensuses, parallelize.py:107-217 has no counterpart - the reference never splits a cluster's
 * consensus) same as ngsid_poa_consensus, but sequence i of `reads` stands for weight[i] reads (host array over the reads of the set, 0 counts as 1):
 * qualities are ignored and every base of the sequence weighs weight[i] (capped at 2^20 per base, like the tile consensuses of the upper levels). */
int32_t ngsid_poa_consensus_weighted(ngsid_ctx* ctx, const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                     const ngsid_poa_params_t* prm, const uint32_t* weight,
                                     uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed);

typedef struct {
    int32_t iters;            /* --racon_iter (NGSpeciesID:212) */
    int32_t window;           /* racon -w 500 */
    double  quality_threshold;/* racon -q 10 */
    double  error_threshold;  /* racon -e 0.3 */
    int32_t match, mismatch, gap;   /* racon -m 3 -x -5 -g -4 */
    int32_t k, w;             /* minimizer parameters used for strand detection (replaces minimap2 -x map-ont) */
    int32_t tile_depth, band, node_cap;
    int32_t aln_match, aln_mismatch, aln_open, aln_ext;  /* read->backbone aligner (replaces the edlib NW path of racon) */
    int32_t trim;             /* 0 none; 1 = racon: coverage-trim the consensus of TGS windows (mean read length > 1000); 2 = trim every window AND every tile consensus of the
                                 hierarchy (+ the one-third rule on its upper levels): the shipped default; 3 (round 5) = tiles as in 2, windows as in 1: a window keeps the ends
                                 of its backbone where no layer reaches them - what aln_mode 3 needs, whose clipped layers stop short of the backbone's ends */
    int32_t aln_mode;         /* read->backbone aligner: 0 = semi-global affine (aln_* scores, all end gaps free);
                                 1 = unit-cost edit distance, read end to end, backbone ends free (what racon gets from edlib inside the
                                 minimap2 span); traceback prefers match/mismatch, then a read-only column, then a backbone-only column;
                                 end column = leftmost minimum of the last row; non-ACGT letters match nothing;
                                 2 = the library's default (currently mode 1);
                                 3 = mode 1 + OVERLAP-SPAN CLIPPING (round 5): of the alignment only the columns from the first to the last run of at least 15 equal
                                 columns count - read ends that do not align stay out of the layers, as minimap2's q_begin / q_end keep them out of racon's edlib call
                                 (consensus.py:121: PAF without CIGAR = first anchor to last anchor of the chain, k = 15 for -x map-ont; racon src/overlap.cpp
                                 find_breaking_points aligns only that span).  A read without such a run contributes nothing.  The CLI uses it where read ends are KNOWN
                                 to overhang the backbone: polishing primer-trimmed drafts (--primer_file / --remove_universal_tails) */
    int32_t stop_when_stable; /* 1 = a group whose backbone comes back unchanged from an iteration is not polished again: the polisher is a
                                 deterministic function of (backbone, reads), so every further iteration would return the same string and
                                 the same n_used - the result is identical, only the time differs.  0 = always run `iters` iterations */
    int32_t single_below;     /* round 6: a polishing window with FEWER layers than this is built as ONE graph (racon's own order: backbone, then the layers by first
                                 position), see ngsid_poa_params_t.single_below; 0 = off */
} ngsid_polish_params_t;

/* (a16,a17) replaces run_racon's (minimap2 -> racon) x racon_iter chain (consensus.py:107-126).
 * backbones: one sequence per group (qual ignored); reads grouped like ngsid_poa_consensus; a read may be listed under ONE group only
 * (NGSID_ERR_ARG otherwise: orientation and window layers are kept per read).
 * n_used[g] (may be NULL) = reads that contributed at least one window layer in the last iteration. */
int32_t ngsid_polish(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                     const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                     uint64_t* out_off, uint8_t* out, uint64_t out_cap, uint64_t* needed, uint64_t* n_used);

/* (a17, boundary 8b(1)) ngsid_polish with the sequence after EVERY iteration = the racon_polished_it_{i}.fasta files run_racon leaves behind
 * (consensus.py:112-120).  it_off has iters * n_groups + 1 entries; entry it * n_groups + g is group g after iteration it (the last iteration is
 * what ngsid_polish returns); it_used likewise (may be NULL).  With stop_when_stable the iterations after a group became stable repeat its string. */
int32_t ngsid_polish_trace(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                           const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                           uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used);
/* (a17, boundary 8b(1)) ngsid_polish_trace + what `minimap2 -x map-ont center reads` leaves in read_alignments_it_{i}.paf (consensus.py:112-121): the read -> backbone
 * alignment of every LISTED read x (x < n_listed = grp_off[n_groups]) in every iteration, it_aln[(it * n_listed + x) * 6 ...] = { strand (0 '+', 1 '-', -1 = no alignment: no
 * PAF line), q_begin, q_end, t_begin, t_end, distance } in PAF coordinates (0-based, end exclusive, the query interval on the read's ORIGINAL strand); distance = unit-cost edit
 * distance of the whole read against the backbone with free backbone ends (-1 with aln_mode 0).  A group that is stable (stop_when_stable) repeats its records. */
int32_t ngsid_polish_trace_aln(ngsid_ctx* ctx, const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                               const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                               uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used, int32_t* it_aln);

/* (8e step 3, boundary 8b) the merge rounds of parallel_clustering (parallelize.py:169-217) on the all-gathered representatives of `n_batches`
 * shards: reps = host read set with qualities, batch[i] = 1-based shard index of representative i, score / hpc_err / acc_rank as in
 * ngsid_cluster_greedy.  rep_of[i] = index of the final representative of i.  Every round's clustering is ngsid_cluster_greedy; the schedule
 * (who meets whom, who keeps its database) is include/ngsid_merge_schedule.h.  Deterministic: every rank computes the same map. */
int32_t ngsid_merge_representatives(ngsid_ctx* ctx, const ngsid_reads_t* reps, const ngsid_cluster_params_t* prm, const uint32_t* acc_rank,
                                    const double* score, const double* hpc_err, const int32_t* batch, int32_t n_batches, int32_t* rep_of);

/* (f2) Host-side ingest / egress helpers of the drop-in CLI (no device work, multi-threaded; csrc/host_io.hip).  They replace the per-record
 * Python of readfq (help_functions.py:13-42) and of the writers (get_sorted_fastq_for_cluster.py:174-177, NGSpeciesID:99-120,
 * consensus.py:203-215).
 * ngsid_host_fastq_index: index of a plain 4-line FASTQ in memory; call with rec == NULL to count (*n_records), then with arrays of that size:
 *   rec[4r..4r+3] = offsets of name (after '@'), sequence, '+' line, quality; returns 1 when the buffer is not a plain 4-line FASTQ (the caller
 *   falls back to the general reader).
 * ngsid_host_gather: dst[dst_off[i] .. +len[i]) = src[src_off[i] .. +len[i]).
 * ngsid_host_normalize_bases: in place, a..z -> A..Z, then anything outside ACGTN -> N; *changed = bytes altered.
 * ngsid_host_write_records: kind 0 = FASTQ records "@name sfx \n seq \n+\n qual \n", kind 1 = TSV lines "pre \t name \n" for reads idx[0..n); sfx / pre are
 *   CSR strings per OUTPUT record (sfx_off NULL = none; per READ, i.e. indexed by idx[j], when sfx_by_read != 0); first_token != 0 cuts the name at the first blank (consensus.py:213). */
/* ngsid_host_thread_cap: upper bound of the worker threads the helpers below start when called from the CALLING thread (0 = none; default: the CPUs the process
 * may use - hardware threads, affinity mask, container quota - at most 32).  Returns the previous bound.  The CLI's background writers take 4 each. */
int32_t ngsid_host_thread_cap(int32_t n);
/* ngsid_host_write_paf: read_alignments_it_{i}.paf of run_racon (consensus.py:112-121), one 12-column PAF line per record of ngsid_polish_trace_aln with strand >= 0: query name
 *   (first token of name + per-read suffix), query length off[i+1]-off[i], q_begin, q_end, strand, tname, tlen, t_begin, t_end, matches (block - distance), block length (the longer
 *   span), 255.  job == NULL writes synchronously, else the call is queued on the background writers (ngsid_host_async_wait). */
int32_t ngsid_host_write_paf(const char* path, uint64_t n, const uint64_t* idx, const uint8_t* names, const uint64_t* name_off, const uint32_t* name_len,
                             const uint8_t* sfx, const uint64_t* sfx_off, const uint64_t* off, const int32_t* aln, const char* tname, uint32_t tlen, int32_t max_threads, uint64_t* job);
int32_t ngsid_host_fastq_index(const uint8_t* buf, uint64_t len, uint64_t* rec, uint32_t* name_len, uint32_t* seq_len, uint64_t cap_records, uint64_t* n_records);
int32_t ngsid_host_gather(const uint8_t* src, const uint64_t* src_off, const uint32_t* len, uint64_t n, uint8_t* dst, const uint64_t* dst_off);
int32_t ngsid_host_normalize_bases(uint8_t* seq, uint64_t len, uint64_t* changed);
/* ngsid_host_count_foreign_bases: how many bytes ngsid_host_normalize_bases would change.
 * ngsid_host_repr_doubles: CPython's repr(float) of every value (shortest round-trip digits, Python's fixed / exponent layout), each preceded by
 *   `prefix` (0 = none), as CSR strings in buf (32 bytes per value must be available) / off[n+1]: the "_score" name suffixes of sorted.fastq
 *   (get_sorted_fastq_for_cluster.py:176). */
int32_t ngsid_host_count_foreign_bases(const uint8_t* seq, uint64_t len, uint64_t* count);
int32_t ngsid_host_repr_doubles(const double* v, uint64_t n, int32_t prefix, uint8_t* buf, uint64_t cap, uint64_t* off, uint64_t* needed);
/* ngsid_host_argsort_desc: order[0..n) = the STABLE argsort of v in descending order (read_array.sort(key=score, reverse=True),
 *   get_sorted_fastq_for_cluster.py:174: equal scores keep their input order; -0.0 == 0.0; NaNs last).
 * ngsid_host_list_positions: pos[i] = number of j < i with rep[j] == rep[i] (rep = the representative's read index, < n): a read's position in its cluster's
 *   list when one clustering pass appends the joining reads in processing order behind the representative (cluster.py:338-345). */
int32_t ngsid_host_argsort_desc(const double* v, uint64_t n, uint64_t* order);
int32_t ngsid_host_list_positions(const int64_t* rep, uint64_t n, int64_t* pos);
/* ngsid_host_group_by_rep: clusters of a representative map (rep[i] = read index of read i's representative, rep[rep[i]] == rep[i]): reps[0..*n_reps) = the
 *   representatives in ascending order, counts[c] / grp_off[c] = size / start of cluster c, order = the reads sorted by cluster, ascending index inside a cluster
 *   (the read lists form_draft_consensus feeds to spoa: representative first, members in processing order; consensus.py:257-266).  reps / counts: room for n
 *   entries, grp_off: n + 1. */
int32_t ngsid_host_group_by_rep(const int64_t* rep, uint64_t n, int64_t* reps, uint64_t* n_reps, uint32_t* order, uint64_t* grp_off, int64_t* counts);
int32_t ngsid_host_group_by_rep32(const int32_t* rep, uint64_t n, int64_t* reps, uint64_t* n_reps, uint32_t* order, uint64_t* grp_off, int64_t* counts);      /* the same on the int32 map ngsid_cluster_greedy returns */
int32_t ngsid_host_write_records(const char* path, int32_t append, int32_t kind, uint64_t n, const uint64_t* idx,
                                 const uint8_t* names, const uint64_t* name_off, const uint32_t* name_len, int32_t first_token,
                                 const uint8_t* sfx, const uint64_t* sfx_off, int32_t sfx_by_read, const uint8_t* seq, const uint8_t* qual, const uint64_t* off);
/* (f2, round 5) the same writer as a BACKGROUND job of the library: returns at once with a job id; eight native worker threads run the jobs, ngsid_host_async_wait(job) blocks until
 * the job is done and returns ITS result (NGSID_ERR_ARG for an id that is unknown or was waited for already).  Every pointer of the call (path excepted: copied) must stay valid
 * until the job was waited for.  Why: a caller with an interpreter lock (the Python CLI) that runs its writers as interpreter threads makes its GPU-launching thread compete for
 * that lock at every return from the library (get_sorted_fastq_for_cluster.py:174-182 and consensus.py:203-215 write synchronously; this is the overlap, without the lock). */
int32_t ngsid_host_write_records_async(const char* path, int32_t append, int32_t kind, uint64_t n, const uint64_t* idx,
                                       const uint8_t* names, const uint64_t* name_off, const uint32_t* name_len, int32_t first_token,
                                       const uint8_t* sfx, const uint64_t* sfx_off, int32_t sfx_by_read, const uint8_t* seq, const uint8_t* qual, const uint64_t* off,
                                       int32_t max_threads /* helper threads of this job; <= 0: 8.  A job that runs beside a launch-bound GPU stage should take few (the CLI: 4 for sorted.fastq) */, uint64_t* job);
int32_t ngsid_host_async_wait(uint64_t job);
/* (f2, round 5) read-only mapping of an input file (an empty file: *data = NULL, *len = 0) and its release; in_background != 0: a detached native thread unmaps (gigabytes take
 * tens of milliseconds, which a caller in the middle of its ingest - help_functions.readfq reads the file line by line instead - need not wait for) */
int32_t ngsid_host_map_file(const char* path, const uint8_t** data, uint64_t* len);
int32_t ngsid_host_unmap_file(const uint8_t* data, uint64_t len, int32_t in_background);
/* decimal strings of n integers as CSR (buf, off[n + 1]); *needed = bytes of buf (cap = 0 sizes): the output ids of final_clusters.tsv (NGSpeciesID:30-50 formats them line by line) */
int32_t ngsid_host_int_prefixes(const int64_t* v, uint64_t n, uint8_t* buf, uint64_t cap, uint64_t* off, uint64_t* needed);

/* (f4) infix ("HW") location of a primer inside a consensus end = edlib.align(primer, window, mode="HW", task="locations", k=max_ed,
 * additionalEqualities=IUPAC)["locations"][0] as barcode_trimmer.find_barcode_locations uses it (barcode_trimmer.py:34-60): *ed = smallest edit
 * distance or -1 when above max_ed, [*start, *end] (inclusive) = the first end position with that distance and the smallest start ending there. */
int32_t ngsid_host_infix_locate(const uint8_t* query, int32_t qlen, const uint8_t* target, int32_t tlen, int32_t max_ed, int32_t iupac,
                                int32_t* ed, int32_t* start, int32_t* end);

/* Scheduling options of a context, for tests and tools: "cluster_block" (reads per speculative block, 0 = adaptive), "ed_band" (Ukkonen band of the
 * polisher's first aligner launch, 0 = off, -1 = automatic), "ed_win_all", "align32" (force the int32 clustering aligner), "align_noclass" (no
 * query-length classes), "align_paired" (0 = the one-pair-per-wave kernel for every length class; default 1: two pairs per wave for every single-strip length class, i.e. queries of up to 896 bases, in batches of at least 4 096 pairs), "poa_tiles_per_cu", "poa_out_slots" (output slots per POA tile of the first attempt, default 4: memory of the level buffers; a tile with more outputs is retried with more), "minimizers_lean" (the long-read LDS layout of the minimizer kernel for every read), "poa_host_levels" (hierarchy levels driven by the host), "scratch_budget_mb" (cap of the aligners' traceback scratch: several contexts on one GPU), "poa_level_budget_mb" (byte budget of the POA hierarchy's level buffers: a call runs in batches of whole units under it; default a third of the free device memory per live context, 1 - 48 GB), "minimizers_chunk_bases" (reads per launch of the minimizer kernel, by bases; default 256 M), "touch" (one trivial device operation), "release_scratch" (frees the context's grow-only scratch and the block cache now).  RESULTS NEVER
 * DEPEND ON THEM (tests/test_gpu_stress.py runs the parity suites under several settings); the library reads no environment variable for them.
 * Environment variables the library does read, none of which changes a result: NGSID_HOST_THREADS (thread count of the ngsid_host_* helpers,
 * default = hardware threads, at most 32), and three developer aids - NGSID_DEBUG_SYNC (synchronise and log after every launch),
 * NGSID_POA_PHASES (cycle counters of the POA kernel on stderr), NGSID_HOST_TIMERS (host-side section timers on stderr). */
int32_t ngsid_ctx_option(ngsid_ctx* ctx, const char* name, int64_t value);

/* (b, f2) A caller that makes several calls on the SAME reads (the CLI: cluster + draft consensus + polish; NGSpeciesID:88-131 keeps one
 * sorted read list for all stages) copies them to HBM once: *dev receives a read set with mem = NGSID_MEM_DEVICE whose buffers belong to the
 * library until ngsid_reads_release (or ngsid_destroy of the last context).  host->qual may be NULL.  The device read set is valid for
 * every context on the same device.  Release it when no call of ANY context that uses it is in flight: the release waits for the streams of `ctx` only
 * (round 6: contexts that work side by side on one device do not wait for each other's kernels). */
int32_t ngsid_reads_upload(ngsid_ctx* ctx, const ngsid_reads_t* host, ngsid_reads_t* dev);
int32_t ngsid_reads_release(ngsid_ctx* ctx, ngsid_reads_t* dev);
/* (b, f2) the reads idx[0..n) (host array) of a device-resident read set, in that order, as a NEW device-resident read set (released like the one of ngsid_reads_upload).
 * The CLI uploads the reads once in file order, scores them there and takes the score order (get_sorted_fastq_for_cluster.py:174) as a gather on the device.
 * *foreign (may be NULL) = bases of the result outside A/C/G/T/N, what ngsid_host_count_foreign_bases counts on the host. */
int32_t ngsid_reads_subset(ngsid_ctx* ctx, const ngsid_reads_t* dev_in, const uint64_t* idx, uint64_t n, ngsid_reads_t* dev_out, uint64_t* foreign);

/* Measurement hooks (bench.py): when enabled every kernel launch of this ctx is bracketed by HIP events on the
 * ctx's own stream; ngsid_profile_read synchronises and writes "kernel_name launches total_ms\n" lines (and resets). */
int32_t ngsid_profile_enable(ngsid_ctx* ctx, int32_t on);
int32_t ngsid_profile_read(ngsid_ctx* ctx, char* buf, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
