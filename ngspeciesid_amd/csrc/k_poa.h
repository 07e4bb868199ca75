// k_poa.h - shared declarations of the POA tile engine (device job description + host launcher)
#pragma once
#include "ngsid_internal.h"

struct PSeq { const uint8_t* s; const uint8_t* q; int32_t len, uw; uint32_t cw; int32_t mode, a0, a1; };   // 40 bytes

struct PoaJobSet {
    const PSeq* seqs; const PSeq* bbs; const uint32_t* seq_idx; const uint32_t* job_off; const int32_t* job_bb; uint32_t njobs; const uint32_t* job_list; uint32_t nrun;   /* job_list != null: this launch runs tiles job_list[0..nrun) (band-edge redo); else all njobs */
    const uint8_t* job_final;                 /* != null (round 5, polish trim 3): job_final[j] != 0 = tile j ends its unit: its consensus is NOT trimmed (trim_tiles counts as 0) */
    const uint32_t* nrun_dev;                 /* != null: the number of tiles to run is read from device memory (device-driven hierarchy: no host round trip between levels) */
    int m, n, g, Vcap, Ecap, Lmax, D, node_cap, trim_tiles;       // D = output slots per job
    int32_t* Hglob; uint8_t* dirglob; uint32_t* covglob;           // per resident workgroup scratch (filled by poa_run_jobs)
    uint8_t* out; int32_t* out_len; int32_t* out_span /* (a0, a1) per slot, may be null */; uint64_t* out_cw; uint32_t* out_n; uint32_t* out_cov; uint32_t* dropped; uint32_t* slot_overflow; unsigned long long* phase_cycles; int phase_detail; unsigned long long* stat_rows /* != null (profiling on): DP rows are added here, one atomic per tile */;   // optional dev instrumentation (NGSID_POA_PHASES=1: phase cycles; =2: also row kinds / checksums, which cost extra passes)
};

// Tiles of a level: n sequences in tiles of D in order; a remainder of fewer than (D + 1) / 2 sequences does not get a tile of its own but joins the last full tile
// (round 3: the one- and two-member remainder tiles were the weak spot of the hierarchy - a single heavy minority member can carry its insertions through them).
// Tile t covers [t D, t == ntiles - 1 ? n : (t + 1) D).  Mirrors oracle/ngsid_oracle_poa.c: ntiles_of.
__host__ __device__ inline uint32_t poa_ntiles(uint32_t n, uint32_t D) { if (n == 0) return 0; if (D == 0 || n <= D) return 1; const uint32_t r = n % D; return (r != 0 && r < (D + 1) / 2) ? n / D : (n + D - 1) / D; }

size_t poa_lds_bytes(int Vc, int Ec, int Lm, int BW);
int32_t poa_run_jobs(ngsid_ctx* ctx, PoaJobSet J, int band);
// device-driven hierarchy: scratch for all three band instances sized once (poa_prepare), then launches that neither allocate nor touch the host
struct PoaPlan { int Vcap, Ecap, Lmax; uint32_t nwg_main, nwg_redo; int band0; };
int32_t poa_prepare(ngsid_ctx* ctx, PoaPlan& P, uint32_t max_jobs);
int32_t poa_launch(ngsid_ctx* ctx, const PoaPlan& P, PoaJobSet J, int band, bool redo, uint32_t* work_ctr);
