/* ngsid_oracle.h - CPU restatement ("oracle") of the NGSpeciesID hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (libngsid_hip.so / ngspeciesid_amd) never links, imports or calls it.
 *
 * Every ongsid_* function has the signature of its ngsid_* counterpart in include/ngsid.h minus the
 * ctx argument, and restates the reference algorithm sequentially (citations in ngsid_oracle.c).
 *
 * PARITY PINNING (see DESIGN.md "Oracle"):
 *   pinned against the reference's own Python, imported in the build container (oracle/make_golden.py):
 *     HPC, minimizers, HPC error rate, p-table selection, get_all_hits/get_best_cluster,
 *     reads_to_clusters end-to-end (with this oracle's aligner behind a parasail-shaped shim),
 *     batch_list / parallel_clustering tree merge, cigar_to_seq window identity, read scoring.
 *   PARITY UNPINNED (third-party native code absent from /root/reference and from this image):
 *     parasail 1.2.4 traceback tie-breaks and wildcard scoring; spoa 4.0.7; racon 1.4.20; minimap2 2.23.
 *     Their published algorithms are restated from documentation/memory and anchored on the
 *     reference's call sites (cluster.py:131-135, consensus.py:59-63,87,121-122) and on synthetic
 *     ground truth (consensus == generating amplicon).
 */
#ifndef NGSID_ORACLE_H
#define NGSID_ORACLE_H
#include "../include/ngsid.h"
#ifdef __cplusplus
extern "C" {
#endif

uint32_t ongsid_abi_version(void);
const char* ongsid_last_error(void);

int32_t ongsid_score_reads(const ngsid_reads_t* reads, int32_t k, double q_threshold,
                           double* score, double* err_rate, uint8_t* keep);
int32_t ongsid_hpc_minimizers(const ngsid_reads_t* reads, int32_t k, int32_t w,
                              uint64_t* mz_off, uint64_t* codes, uint32_t* pos, uint64_t cap, uint64_t* needed,
                              uint32_t* hpc_len, double* hpc_err);
int32_t ongsid_cluster_greedy(const ngsid_reads_t* reads, const ngsid_cluster_params_t* prm,
                              const uint32_t* acc_rank, const int32_t* prev_batch, const double* known_err,
                              int32_t* rep_of_read, double* hpc_err_out, uint8_t* status_out, uint64_t counters[4]);
int32_t ongsid_ed_align_batch(const ngsid_reads_t* queries, const ngsid_reads_t* targets, const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                              int32_t window, int32_t bp_windows, int32_t* distance, int32_t* span, int32_t* bp);
int32_t ongsid_sg_align_batch(const ngsid_reads_t* queries, const ngsid_reads_t* targets,
                              const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                              int32_t match, int32_t mismatch, const int32_t* open, int32_t ext,
                              int32_t k, const int32_t* match_id,
                              int32_t* score, int32_t* n_cols, int32_t* n_match, int32_t* region);
/* single pair with CIGAR text (=XID, end gaps included) for the parasail-shaped shim */
int32_t ongsid_host_infix_locate(const uint8_t* query, int32_t qlen, const uint8_t* target, int32_t tlen, int32_t max_ed, int32_t iupac, int32_t* ed, int32_t* start, int32_t* end);
int32_t ongsid_merge_representatives(const ngsid_reads_t* reps, const ngsid_cluster_params_t* prm, const uint32_t* acc_rank,
                                     const double* score, const double* hpc_err, const int32_t* batch, int32_t n_batches, int32_t* rep_of);
int32_t ongsid_sg_align_cigar_batch(const ngsid_reads_t* queries, const ngsid_reads_t* targets, const uint32_t* q_idx, const uint32_t* t_idx, uint64_t n_pairs,
                                    int32_t match, int32_t mismatch, const int32_t* open, int32_t ext,
                                    int32_t* score, uint64_t* ops_off, uint8_t* ops, uint64_t cap, uint64_t* needed);
int32_t ongsid_sg_align_cigar(const uint8_t* q, int32_t n, const uint8_t* t, int32_t m,
                              int32_t match, int32_t mismatch, int32_t open, int32_t ext,
                              char* cigar, int32_t cigar_cap, int32_t* score);
int32_t ongsid_poa_consensus(const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                             const ngsid_poa_params_t* prm,
                             uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed);
int32_t ongsid_poa_consensus_cov(const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                 const ngsid_poa_params_t* prm, uint64_t* cons_off, uint8_t* cons, uint32_t* cov, uint64_t cons_cap, uint64_t* needed);
int32_t ongsid_poa_consensus_weighted(const ngsid_reads_t* reads, const uint32_t* read_order, const uint64_t* grp_off, uint64_t n_groups,
                                      const ngsid_poa_params_t* prm, const uint32_t* weight, uint64_t* cons_off, uint8_t* cons, uint64_t cons_cap, uint64_t* needed);
int32_t ongsid_polish(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                      const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                      uint64_t* out_off, uint8_t* out, uint64_t out_cap, uint64_t* needed, uint64_t* n_used);

int32_t ongsid_polish_trace(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                            const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                            uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used);
int32_t ongsid_polish_trace_aln(const ngsid_reads_t* backbones, const ngsid_reads_t* reads, const uint32_t* read_order,
                                const uint64_t* grp_off, uint64_t n_groups, const ngsid_polish_params_t* prm,
                                uint64_t* it_off, uint8_t* it_out, uint64_t it_cap, uint64_t* needed, uint64_t* it_used, int32_t* it_aln);
/* tile engine of the consensus / polishing oracle: 0 = node-indexed graph (ngsid_oracle_poa.c, the definition), 1 = rank-ordered graph
   (ngsid_oracle_poa_rank.c, the representation of csrc/k_poa.hip); e < 0 only reads.  Returns the previous value. */
int32_t ongsid_debug_poa_engine(int32_t e);

/* tie-break envelope of the aligner (ngsid_oracle.c: g_sg_tiebreak); mode < 0 only reads.  Returns the previous mode. */
int32_t ongsid_debug_sg_tiebreak(int32_t mode);
/* round 5, reference-order experiments of the polisher (oracle only): bit 0 = overlap-span clipping of the read -> backbone alignment (what minimap2's q_begin / q_end do before
   racon's edlib call), bit 1 = sub-graph alignment of layers that do not span their window (racon src/window.cpp); returns the previous value, r < 0 only reads */
int32_t ongsid_debug_polish_rules(int32_t r);

/* debugging taps used by the golden tests (per-read mapping-stage triple of cluster.py:302) */
int32_t ongsid_debug_enable_trace(int32_t* best_m, int32_t* nshared, double* ratio, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif
