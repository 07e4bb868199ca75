#!/usr/bin/env python3
"""bench.py - reads/s end-to-end (cluster + spoa-style consensus + racon-style polish x3) on synthetic 750 bp ONT reads.

    python bench.py --gpus 1 --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One step = one pass of the hot path over the whole batch of synthetic reads, inputs resident in HBM.
Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how roofline / cpu_baseline are obtained.
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np
import torch


# BASELINE.json configurations: species, read length, quality profile, (k, w), abundance vector and --abundance_ratio.
# `reads` = reads PER GPU in the default weak-scaling mode (C4 / C5 are quoted on 8 GPUs: 10 M / 8 and 2 M / 8), and `total` = the size of the
# ONE global set of --scaling strong.  C1 (sample_h1 through the CLI) is a test, not a bench workload.
CONFIGS = {
    "c2": dict(preset="--ont", reads=100000, total=100000, species=1, length=750, mu=17.0, k=13, w=20, abundance_ratio=0.1, geometric=None, flags="--ont --abundance_ratio 0.1"),
    "c3": dict(preset="--ont", reads=1000000, total=1000000, species=5, length=750, mu=17.0, k=13, w=20, abundance_ratio=0.02, geometric=None, flags="--ont --abundance_ratio 0.02"),
    "c4": dict(preset="--ont", reads=1250000, total=10000000, species=50, length=750, mu=17.0, k=13, w=20, abundance_ratio=0.005, geometric=None, flags="--ont --abundance_ratio 0.005"),
    "c5": dict(preset="--isoseq", reads=250000, total=2000000, species=20, length=2000, mu=30.0, k=15, w=50, abundance_ratio=0.002, geometric=0.8, flags="--isoseq --abundance_ratio 0.002"),
}


def gen_sorted_reads(api, n_reads, n_species, L, mu, seed, device, abundance=None, rc_fraction=0.0, k=13, rng="torch"):
    """synthetic reads, scored (f1) and physically ordered by score descending (stable) = the greedy order."""
    from ngspeciesid_amd import synth
    from ngspeciesid_amd._capi import ReadSet
    tr = (lambda m: (sys.stderr.write("[gen pid %d] %s\n" % (os.getpid(), m)), sys.stderr.flush())) if os.environ.get("NGSID_BENCH_TRACE") else (lambda m: None)
    sp = synth.make_species(n_species, L, 0.15, seed=1)
    tr("species made")
    rd = synth.make_reads(sp, n_reads, mu=mu, seed=seed, device=device, abundance=abundance, rc_fraction=rc_fraction, rng=rng)
    rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
    torch.cuda.synchronize(device)       # the library runs on its own HIP stream: torch's generator kernels must have finished writing the reads
    tr("reads made")
    score, err, keep = api.score_reads(rs, k, 7.0)
    tr("reads scored")
    keep_idx = np.nonzero(keep)[0]
    perm = keep_idx[np.argsort(-score[keep_idx], kind="stable")]
    off = rd["off"]; lens = (off[1:] - off[:-1])
    perm_t = torch.from_numpy(perm).to(device)
    nlen = lens[perm_t]
    noff = torch.zeros(len(perm) + 1, dtype=torch.int64, device=device); noff[1:] = torch.cumsum(nlen, 0)
    total = int(noff[-1].item())
    nseq = torch.empty(total, dtype=torch.uint8, device=device); nqual = torch.empty(total, dtype=torch.uint8, device=device)
    CH = 1 << 17
    for a in range(0, len(perm), CH):
        b = min(len(perm), a + CH)
        l = nlen[a:b]; src0 = off[perm_t[a:b]]; dst0 = noff[a:b]
        idx = torch.arange(int(l.sum().item()), device=device) - torch.repeat_interleave(dst0 - dst0[0], l)
        src = idx + torch.repeat_interleave(src0, l)
        d0 = int(dst0[0].item())
        nseq[d0:d0 + len(src)] = rd["seq"][src]; nqual[d0:d0 + len(src)] = rd["qual"][src]
    out = dict(seq=nseq, qual=nqual, off=noff, species=rd["species"][perm_t], score=score[perm], orig=perm)
    torch.cuda.synchronize(device)       # same for the gather into score order
    del rd, rs, src, idx; torch.cuda.empty_cache()        # the unsorted copy (+ torch's cached generator temporaries: 3 x the read set at 10 M reads) goes back to the driver
    return sp, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=2)      # (two: the first timed step after ONE warm-up step still meets first-time allocations of the pool)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="BASELINE.json configuration: c2 100 k x 750 bp 1 species | c3 (default, the one `metric` is quoted on) 1 M x 750 bp 5 species | "
                    "c4 10 M x 750 bp 50 species over 8 GPUs (1.25 M per GPU) | c5 2 M x 2 kb CCS 20 species, geometric abundance, k15/w50 over 8 GPUs (250 k per GPU)")
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (weak scaling) or in total (strong scaling); default: the configuration's")
    ap.add_argument("--species", type=int, default=None)
    ap.add_argument("--length", type=int, default=None)
    ap.add_argument("--mu", type=float, default=None)
    ap.add_argument("--tile-depth", type=int, default=4, help="reads per POA tile (pipeline / CLI default 4 since round 5)")
    ap.add_argument("--band", type=int, default=0, help="POA band width in columns of the first attempt (64 / 128 / 256); 0 = library default (64 for reads up to 3 000 bases)")
    ap.add_argument("--node-cap", type=int, default=0, help="POA graph capacity in 1/16 of the first sequence length (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=1500, help="reads per worker process of the cpu_baseline leg")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the cpu_baseline leg (0 = all host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-step", action="store_true", help="skip the extra step with stop_when_stable (profiling runs that count per-step traffic)")
    ap.add_argument("--no-cli", action="store_true", help="skip the file-in -> files-out leg (the drop-in CLI on the same reads, reported as config.cli)")
    ap.add_argument("--cli-t", type=int, default=1, help="--t of the CLI leg (the reference's batch count; 1 = one clustering pass)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong for --config c4 / c5 with --gpus N > 1 (BASELINE quotes them as ONE global set over 8 GPUs), weak otherwise.  weak: --reads per GPU, an independent set per rank.  strong: ONE global score-sorted set of --reads reads, rank g gets batch g+1 of the "
                         "reference's `--t N` partition (parallelize.batch_list total_nt), so the N-GPU membership is the reference's --t N membership of that set")
    ap.add_argument("--check-membership", action="store_true", help="strong scaling: after the timed region rank 0 replays the `--t N` schedule on one GPU and compares the membership")
    ap.add_argument("--master-port", type=int, default=0, help="--gpus N > 1 started directly: the rendezvous port of the ranks this process launches (0 = a free one)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` IS the N-GPU run: this process becomes `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>`,
        # one rank per GPU over RCCL (backend nccl; NGSID_DIST_BACKEND=gloo only as a dev aid for more ranks than GPUs).  stdout is inherited, so rank 0's JSON line is this command's.
        port = args.master_port
        if not port:
            import socket
            with socket.socket() as s_: s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
        if os.environ.get("NGSID_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < args.gpus:
            sys.exit("bench.py --gpus %d: this node shows %d GPU(s); one RCCL rank per GPU is required (NGSID_DIST_BACKEND=gloo lets ranks share a GPU, as a dev aid)" % (args.gpus, torch.cuda.device_count()))
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:])
    cfg = CONFIGS[args.config]
    if args.scaling is None:      # the BASELINE shape of the configuration: C4 / C5 are ONE global set over the GPUs of a node (strong); C2 / C3 are per-GPU workloads (weak)
        args.scaling = "strong" if (args.config in ("c4", "c5") and args.gpus > 1) else "weak"
    if args.reads is None: args.reads = int(os.environ.get("NGSID_BENCH_READS", cfg["total"] if args.scaling == "strong" else cfg["reads"]))
    if args.species is None: args.species = cfg["species"]
    if args.length is None: args.length = cfg["length"]
    if args.mu is None: args.mu = cfg["mu"]
    K_, W_, AB_ = cfg["k"], cfg["w"], cfg["abundance_ratio"]
    abundance = [cfg["geometric"] ** i for i in range(args.species)] if cfg["geometric"] else None

    # stdout carries exactly ONE line (the JSON): everything native code prints to fd 1 (RCCL's banner, rocm warnings) goes to stderr instead
    sys.stdout.flush(); json_fd = os.dup(1); os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or os.environ.get("NGSID_FORCE_DIST") == "1", "--gpus %d but the launcher started %d rank(s): n_gpus of the line would not be what ran" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path)"
    ndev = torch.cuda.device_count()
    local = local % ndev                      # (dev aid: more ranks than GPUs only with NGSID_DIST_BACKEND=gloo, e.g. 2 ranks on a 1-GPU box)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    backend = os.environ.get("NGSID_DIST_BACKEND", "nccl")
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    force_dist = os.environ.get("NGSID_FORCE_DIST") == "1"          # dev aid: run the sharded code path (and its RCCL collectives) with a single rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    _t00 = time.perf_counter()
    def trace(msg):           # dev aid: NGSID_BENCH_TRACE=1 -> per-rank progress on stderr
        if os.environ.get("NGSID_BENCH_TRACE"): sys.stderr.write("[bench rank %d %.1fs] %s\n" % (rank, time.perf_counter() - _t00, msg)); sys.stderr.flush()
    from ngspeciesid_amd import runtime, pipeline
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.ptable import select_p_table
    trace("process group up, creating the context")
    api = runtime.get_api(local)
    trace("context created")
    if world > ndev:      # dev aid (ranks sharing one GPU): the aligners' scratch budget is sized for a GPU of one's own by default
        api.set_option("scratch_budget_mb", max(2048, 32768 // world))
        if not os.environ.get("NGSID_LANES_FORCE"): api.lanes = 1     # ... and the ranks already overlap each other on the device: no lanes below them (each would bring a second context's scratch; NGSID_LANES_FORCE=1 keeps them: the test of lanes under ranks on a one-GPU box)
    ptab = select_p_table(K_, W_)
    # multi-process runs draw their reads from plain integer tensor arithmetic (synth._HashRng) instead of torch generators: eight processes sharing one GPU were seen to
    # stall inside torch's generator kernels; the one-GPU workload keeps the torch generator = the read set of every earlier round
    RNG_ = os.environ.get("NGSID_BENCH_RNG", "hash" if world > 1 else "torch")
    rd_global = None
    if args.scaling == "strong" and (world > 1 or force_dist):
        # every rank builds the same global set (same seed, same device type) and keeps its own `--t N` batch of it
        from ngspeciesid_amd import parallelize
        sp, rd_global = gen_sorted_reads(api, args.reads, args.species, args.length, args.mu, seed=7, device=dev, abundance=abundance, k=K_, rng=RNG_)
        goff = rd_global["off"]; glens = (goff[1:] - goff[:-1]).cpu().numpy()
        batches = parallelize.batch_list_total_nt(glens, world)
        a, b = batches[rank] if rank < len(batches) else (len(glens), len(glens))
        o0, o1 = int(goff[a].item()), int(goff[b].item())
        rd = dict(seq=rd_global["seq"][o0:o1].clone(), qual=rd_global["qual"][o0:o1].clone(), off=(goff[a:b + 1] - goff[a]).clone(), species=rd_global["species"][a:b],
                  score=rd_global["score"][a:b], orig=rd_global["orig"][a:b])
        shard_start = a
        if not (args.check_membership and rank == 0): rd_global = None
    else:
        sp, rd = gen_sorted_reads(api, args.reads, args.species, args.length, args.mu, seed=7 + rank, device=dev, abundance=abundance, k=K_, rng=RNG_)
    torch.cuda.synchronize()
    rs = ReadSet.from_torch(rd["seq"], rd["qual"], rd["off"])
    n = rs.n
    trace("reads ready: %d local" % n)
    acc_rank = np.asarray(rd["orig"], dtype=np.uint32)            # stand-in for the accession order (unique, deterministic)
    # The headline runs ALL three polishing iterations (stop_when_stable off).  The library's default stops polishing a cluster once an
    # iteration returns its backbone unchanged (identical result, less work); that rate is reported separately as `with_stable_stop`.
    kw = dict(k=K_, w=W_, abundance_ratio=AB_, racon_iter=3, tile_depth=args.tile_depth, band=args.band, p_shared=ptab, polish_stop_when_stable=False)
    if args.node_cap and world == 1: kw["node_cap"] = args.node_cap

    def step(T=None, **over):
        if dist is None:
            return pipeline.run_hot_path(api, rs, rd["score"], acc_rank=acc_rank, timings=T, **dict(kw, **over))
        from ngspeciesid_amd import distributed
        r = distributed.sharded_hot_path(api, rs, rd["score"], acc_rank_local=acc_rank, timings=T, device=comm_dev, **dict(kw, **over))
        # same result shape as the single-GPU path for the checks below
        r["rep_of"] = r["final_gid"]; r["centers"] = [(c[0], c[1], c[2], c[3], []) for c in r["centers"]]
        return r

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    trace("warm-up done")
    import ctypes as C
    api.profile_enable(not os.environ.get("NGSID_BENCH_NOPROF"))      # (this context and its lane contexts; dev aid: NGSID_BENCH_NOPROF=1 times the steps without the per-launch HIP events)
    T = {}
    barrier(); t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step(T)
    barrier(); dt = time.perf_counter() - t0
    trace("timed region done: %.2f s, stages %s" % (dt, {k_: round(v, 2) for k_, v in T.items()}))
    if os.environ.get("NGSID_BENCH_TRACE"):
        for f_ in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.stat"):
            try: trace("%s: %s" % (f_, open(f_).read().replace("\n", " ")))
            except Exception: pass
    kern, kern_per_ctx = api.profile_read()          # summed over the lane contexts (config.lanes): the host_<api> lines are wall clocks of calls that run side by side
    api.profile_enable(False)
    n_lanes = len(kern_per_ctx)
    for nm_ in ("hbm_peak_bytes", "hbm_live_bytes"):          # (process-wide figures: every context reports the same one)
        if nm_ in kern: kern[nm_] = max((d_.get(nm_, (0, 0.0)) for d_ in kern_per_ctx), key=lambda v_: v_[0])
    # one extra (untimed for `value`) step with the library default: polishing of a cluster stops once an iteration leaves it unchanged
    res_stop, dt_stop = None, 0.0
    if not args.no_extra_step:
        barrier(); t1 = time.perf_counter()
        res_stop = step(polish_stop_when_stable=True)
        barrier(); dt_stop = time.perf_counter() - t1
    # with lanes (config.lanes) the launches of two contexts overlap on the device and the HIP-event bracket of a launch includes what it waited for the other lane: one more
    # untimed step in ONE context gives every kernel's time alone (roofline.one_lane: the figures the committed counter passes and rocprof summaries of a single lane match)
    one_lane = None
    if n_lanes > 1 and not args.no_extra_step:
        saved_ = api.lanes; api.lanes = 1; api.profile_enable(True)
        try: step()
        finally: api.lanes = saved_
        k1_, _ = api.profile_read(); api.profile_enable(False)
        one_lane = {nm_: v_[1] for nm_, v_ in k1_.items() if nm_.startswith("k_")}
    ranks_seen = 1; stage_max = None
    if dist is not None:
        t = torch.tensor([dt], device=comm_dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t.item())
        ones = torch.ones(1, device=comm_dev, dtype=torch.int64); dist.all_reduce(ones); ranks_seen = int(ones.item())        # how many ranks took part in the collectives of this run
        gpus_seen = torch.zeros(64, device=comm_dev, dtype=torch.int64); gpus_seen[local] = 1; dist.all_reduce(gpus_seen); gpus_seen = int((gpus_seen > 0).sum().item())
        names_ = sorted(T); tv = torch.tensor([T[k_] for k_ in names_], device=comm_dev, dtype=torch.float64)
        if len(names_): dist.all_reduce(tv, op=dist.ReduceOp.MAX); stage_max = {k_: float(v) for k_, v in zip(names_, tv.tolist())}
        tn = torch.tensor([n], device=comm_dev, dtype=torch.float64); dist.all_reduce(tn); n_total = int(tn.item())
    else:
        n_total = n
    if dist is not None:
        t = torch.tensor([dt_stop], device=comm_dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt_stop = float(t.item())
    membership_ok = None; single_same = None
    if args.scaling == "strong" and dist is not None and args.check_membership:
        # N-GPU membership against the reference's `--t N` schedule replayed on ONE GPU (rank 0), outside the timed region
        from ngspeciesid_amd import distributed, parallelize
        from ngspeciesid_amd._capi import cluster_params
        from ngspeciesid_amd.hostutil import make_cluster_fn
        starts = [x for x, _ in batches]
        mine = np.asarray([starts[o] + l for o, l in zip(res["final_owner"], res["final_lidx"])], dtype=np.int64)
        allm = distributed.all_gather_obj(dict(a=int(shard_start), final=mine), comm_dev)
        trace("membership gathered")
        if rank == 0:
            final = np.full(len(glens), -1, dtype=np.int64)
            for m_ in allm: final[m_["a"]:m_["a"] + len(m_["final"])] = m_["final"]
            hrs = ReadSet(rd_global["seq"].cpu().numpy(), rd_global["qual"].cpu().numpy(), rd_global["off"].cpu().numpy().astype(np.uint64))
            fn = make_cluster_fn(api, hrs, np.asarray(rd_global["orig"], dtype=np.uint32), cluster_params(k=K_, w=W_, p_shared=ptab))
            rep_ref, _, _ = parallelize.tree_cluster(fn, glens, np.asarray(rd_global["score"]), world)
            membership_ok = bool(np.array_equal(final, rep_ref))
            trace("replayed --t N on one GPU: membership %s" % membership_ok)
            # ... and the sharded consensus against the single-process path on the whole set (one GPU, one clustering pass)
            grs = ReadSet.from_torch(rd_global["seq"], rd_global["qual"], rd_global["off"])
            one = pipeline.run_hot_path(api, grs, rd_global["score"], acc_rank=np.asarray(rd_global["orig"], dtype=np.uint32), **kw)
            single_same = sorted(c[3] for c in one["centers"]) == sorted(c[3] for c in res["centers"])
            trace("single-process pass done")
    if rank != 0:
        return
    stop_same = None if res_stop is None else [c[3] for c in res_stop["centers"]] == [c[3] for c in res["centers"]]
    reads_per_s = n_total * args.steps / dt
    # ---- quality / property checks at full size (size-independent): cluster purity, consensus vs generating amplicon
    spc = rd["species"].cpu().numpy(); rep_of = res["rep_of"]
    big = [c for c in res["centers"]]
    purity = float(sum(np.bincount(spc[rep_of == r]).max() for r in np.unique(rep_of)) / n)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util_seq import edit_distance
    truths = [s.tobytes().decode() for s in sp]
    ed = []; tset = set(truths)
    for c in big:           # exact hits first (50 species x 50 truths x 16 end trims of a row-vectorised edit distance would take minutes)
        ed.append(0 if c[3] in tset else min(min(edit_distance(c[3][a:len(c[3]) - b if b else None], t) for a in range(4) for b in range(4)) for t in truths))
    # ---- roofline of the dominant kernel (HIP-event times on the library's own stream)
    redo_tiles = kern.pop("poa_band_redo_tiles", (0, 0.0))[0]
    mem_parts = {k_[4:]: round(kern.pop(k_)[0] / 1e9, 2) for k_ in [x for x in kern if x.startswith("mem_")]}      # grow-only scratch of the context by purpose (GB)
    hbm_peak = kern.pop("hbm_peak_bytes", (0, 0.0))[0]; hbm_live = kern.pop("hbm_live_bytes", (0, 0.0))[0]      # device memory handed out by the library's allocator (process wide): high-water mark of the timed steps / still held after them
    poa_rows = kern.pop("poa_dp_rows", (0, 0.0))[0]; sg_cells = kern.pop("sg_dp_cells", (0, 0.0))[0]      # work counters of the timed steps (counted by the kernels themselves, ngsid_profile_read)
    # the dominant KERNEL: ngsid_profile_read also carries the library's host-side wall-clock lines (`host_<api>`, round 5) and counters - only `k_*` lines are kernels
    host_lines = {k_: kern.pop(k_) for k_ in [x for x in kern if not x.startswith("k_")]}
    dom = max(kern.items(), key=lambda kv: kv[1][1]) if kern else (None, (0, 0.0))
    f_aln = float(res["counters"][2]) / n
    L = args.length; M = int(round(0.21 * 0.75 * L)) if K_ <= 13 else int(round(0.054 * 0.75 * L))          # minimizers per read (118 at 750 bp k13/w20, 80 at 2 kb k15/w50)
    n_pol = float(sum(c[0] for c in res["centers"]))                                                                # reads that go through the draft POA and the polisher (clusters above the abundance cut-off)
    # algorithmic bytes per STEP (one byte per base and per quality, every stage streams its input once): aligner f_aln x (2L read + 2L representative); POA 2L per read
    # and pass, 1 draft + 3 polishing passes; the polisher's read -> backbone aligner 2L read + 2L backbone per read and iteration; minimizers 2L in + 12M out
    alg_step = {"k_sg_align": f_aln * n * 4 * L, "k_poa_tile": 4.0 * n * 2 * L, "k_ed_align": 3.0 * n_pol * 4 * L, "k_hpc_minimizers": n * (2 * L + 12 * M),
                "k_count_hits": n * (12 * M + 8), "k_decide_map": n * (12 * M + 8), "k_aln_next": n * 8.0}
    tj, tj_file = None, None
    try:    # HBM bytes per step from the committed PMC passes of THIS round (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, tools/r06_profiles.sh): a counter pass
            # cannot run inside this process, so the figure is a measurement of the recorded commit on the recorded workload, not of this run
        tj_file = next(f for f in ("r06_hbm_traffic.json", "r05_hbm_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
        tj = json.load(open(os.path.join(ROOT, "profiles", tj_file)))
        if not (tj.get("workload_reads") == args.reads and tj.get("config", "c3") == args.config and tj.get("tile_depth", args.tile_depth) == args.tile_depth): tj = None
    except Exception:
        tj = None
    def kernel_line(nm):
        cnt, ms = kern[nm]
        alg = alg_step.get(nm, n * 2.0 * L) * args.steps           # bytes over the timed steps
        sec = ms / 1e3
        v = {"launches_per_step": round(cnt / args.steps, 1), "ms_per_step": round(ms / args.steps, 3), "algorithmic_bytes_per_step": int(alg / args.steps),
             "achieved": round(alg / sec / 1e9, 3) if sec > 0 else None, "unit": "GB/s", "frac": round(alg / sec / 8e12, 6) if sec > 0 else None}
        if tj is not None and nm in tj:
            tb = float(tj[nm]["hbm_bytes_per_step"])
            v.update({"traffic_bytes_per_step": int(tb), "traffic_over_algorithmic": round(tb * args.steps / alg, 1) if alg else None,
                      "traffic_GBps": round(tb * args.steps / sec / 1e9, 1) if sec > 0 else None, "traffic_frac_of_peak": round(tb * args.steps / sec / 8e12, 4) if sec > 0 else None})
        return v
    roof = None
    if dom[0]:
        cnt, ms = dom[1]
        alg_bytes_per_launch = alg_step.get(dom[0], n * 2.0 * L) * args.steps / max(cnt, 1)
        avg_s = ms / 1e3 / max(cnt, 1)
        ach = alg_bytes_per_launch / avg_s / 1e9
        traffic = None; traffic_src = None
        if tj is not None and dom[0] in tj and world == 1:
            traffic = int(tj[dom[0]]["hbm_bytes_per_step"] * args.steps / max(cnt, 1))      # per launch, like `achieved`
            traffic_src = {"file": "profiles/" + tj_file, "measured_at_commit": tj.get("commit"), "launches_per_step_then": tj[dom[0]].get("launches_per_step"),
                           "how": "rocprofv3 --pmc FETCH_SIZE and, in a separate run, --pmc WRITE_SIZE over one bench step; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB as MI355X_MICROARCH.md prescribes for gfx950"}
        whole = n * ((10 + 4 * f_aln) * L + 24 * M + 8)            # the path's algorithmic bytes per read x reads
        roof = {"bound": "hbm", "kernel": dom[0], "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_src,
                "traffic_over_algorithmic": round(traffic / alg_bytes_per_launch, 1) if traffic else None,
                "traffic_GBps": round(traffic / avg_s / 1e9, 1) if traffic else None,
                "limited_by": "VALU instruction issue (the DP rows and the graph bookkeeping of one wave per tile), not HBM bandwidth: see `valu_issue` / `dp_kernels` and DESIGN.md section 4",
                "launches": cnt, "avg_launch_ms": round(ms / max(cnt, 1), 4), "algorithmic_bytes_per_launch": int(alg_bytes_per_launch),
                "kernels": {nm: kernel_line(nm) for nm in ("k_poa_tile", "k_sg_align", "k_ed_align", "k_hpc_minimizers") if nm in kern},
                "whole_path": {"algorithmic_bytes_per_read": round(whole / n, 1), "achieved": round(whole * args.steps / dt / 1e9, 3), "unit": "GB/s", "frac": round(whole * args.steps / dt / 8e12, 6),
                               "what": "SURVEY 8d: ((10 + 4 f_aln) L + 24 M + 8) bytes per read x reads per second of `value`"},
                "note": "integer DP kernel, one wave per tile: bound by VALU issue (`valu_issue`: DP rows counted in this run x instructions per row of the committed counter pass / this run's kernel time), so the fraction of the HBM roofline is small by construction (DESIGN.md sections 4 and 10)"}
        # ---- what actually bounds the two DP kernels: VALU issue.  Work (DP rows / cells) is counted by the kernels in THIS run; the instructions per unit of
        #      work come from the committed SQ-counter pass (POA: profiles/r04_pmc_poa_tile.json, with its commit) or the ISA (aligner): a counter pass cannot run
        #      inside this process.  achieved = work x instructions per unit / the kernel's time in this run (HIP events); peak = CUs x 4 SIMDs x clock / 4 cycles.
        prop = torch.cuda.get_device_properties(dev)
        clk = float(getattr(prop, "clock_rate", 2400000)) * 1e3
        peak_issue = prop.multi_processor_count * 4 * clk / 4.0
        # basis of that peak: MEASURED on this GPU - tools/micro/valu_issue.hip -> profiles/r06_valu_rates.txt (round 6: up to 8 waves per SIMD, 8 and 16 independent registers per
        # stream, two mixes at the kernels' own ratios).  v_mov_b32 / v_fma_f32 reach 2.4 - 2.7 cycles per wave64 instruction and SIMD (the guide's 2 cycles, MI355X_MICROARCH.md:52-54, is
        # approached by these two forms only), plain VOP2 integer adds 3.1 - 3.4, and every form these kernels are made of (v_max_i32, v_pk_*_i16, v_*_dpp, v_bfi, v_and_or, v_max3,
        # v_lshl_add) is flat at 4.3 - 4.4 from 4 to 8 waves per SIMD; the aligner-step mix 4.3 - 4.4, the POA-row mix 4.2 - 4.3.  `frac` divides by 4.0 cycles (an upper bound of the
        # peak for these mixes); `frac_at_measured_mix_rate` by 4.3; `frac_at_the_guides_2_cycles` by 2.0 - all three are printed.
        peak_basis = {"cycles_per_wave_instruction_and_simd": 4.0, "measured_mix_rate_cycles": 4.3,
                      "source": "profiles/r06_valu_rates.txt (tools/micro/valu_issue.hip on MI355X, 1 - 8 waves per SIMD: v_mov / v_fma_f32 2.4 - 2.7 cycles, plain VOP2 integer 3.1 - 3.4, DPP / packed i16 / VOP3 forms and both kernel mixes flat at 4.2 - 4.4)",
                      "guide_value": "2 cycles (MI355X_MICROARCH.md:52-54): approached only by v_mov_b32 / v_fma_f32 (2.4 - 2.7)"}
        band_cols = args.band if args.band else (64 if L <= 3000 else 128)          # NGSID_POA_BAND64_MAXLEN (include/ngsid.h)
        views = {}
        if "k_poa_tile" in kern and poa_rows:
            ms_p = kern["k_poa_tile"][1]; v = {"dp_rows": int(poa_rows), "band_columns": band_cols, "kernel_ms": round(ms_p, 2), "gcups": round(poa_rows * band_cols / (ms_p / 1e3) / 1e9, 1)}
            try:
                ij_file = next(f for f in ("r06_pmc_poa_tile.json", "r05_pmc_poa_tile.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
                ij = json.load(open(os.path.join(ROOT, "profiles", ij_file)))
                if ij.get("workload_reads") == args.reads and ij.get("config", "c3") == args.config:
                    wi = poa_rows * ij["valu_per_row"] / (ms_p / 1e3)
                    v.update({"valu_per_row": ij["valu_per_row"], "salu_per_row": ij.get("salu_per_row"), "achieved": round(wi / 1e9, 2), "peak": round(peak_issue / 1e9, 2), "unit": "G wave-instructions/s", "frac": round(wi / peak_issue, 4), "frac_at_measured_mix_rate": round(wi / (peak_issue * 4.0 / 4.3), 4), "frac_at_the_guides_2_cycles": round(wi / (peak_issue * 2.0), 4), "peak_basis": peak_basis,
                              "pipe_busy_by_counters": ij.get("pipe_busy"),
                              "instructions_per_row_source": {"file": "profiles/" + ij_file, "measured_at_commit": ij.get("commit"), "rows_then": ij.get("rows"), "pipe_busy_then": ij.get("pipe_busy")}})
            except Exception:
                pass
            views["k_poa_tile"] = v
        if "k_sg_align" in kern and sg_cells:
            ms_a = kern["k_sg_align"][1] + kern.get("k_sg_align_side", (0, 0.0))[1]
            per_cell, src = 14.4, "ISA count of the step loop of round 3 (378 VALU per step for two pairs of 12-14 rows per lane), DESIGN.md section 4"
            try:            # round 4: VALU instructions per DP cell from the SQ-counter pass of the bench workload (all k_sg_align* dispatches of one step / the cells the kernels counted in it)
                aj_file = next(f for f in ("r06_pmc_sg_align.json", "r05_pmc_sg_align.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
                aj = json.load(open(os.path.join(ROOT, "profiles", aj_file)))
                if aj.get("workload_reads") == args.reads and aj.get("config", "c3") == args.config:
                    per_cell = float(aj["valu_per_cell"]); src = {"file": "profiles/" + aj_file, "measured_at_commit": aj.get("commit"), "cells_then": aj.get("cells"), "pipe_busy_then": aj.get("pipe_busy")}
            except Exception:
                pass
            wi = sg_cells / 64.0 * per_cell / (ms_a / 1e3)
            views["k_sg_align"] = {"dp_cells": int(sg_cells), "kernel_ms": round(ms_a, 2), "gcups": round(sg_cells / (ms_a / 1e3) / 1e9, 1), "valu_per_cell": per_cell, "achieved": round(wi / 1e9, 2), "peak": round(peak_issue / 1e9, 2),
                                   "unit": "G wave-instructions/s", "frac": round(wi / peak_issue, 4), "frac_at_measured_mix_rate": round(wi / (peak_issue * 4.0 / 4.3), 4), "frac_at_the_guides_2_cycles": round(wi / (peak_issue * 2.0), 4), "peak_basis": peak_basis, "instructions_per_cell_source": src}
        if dom[0] in views: roof["valu_issue"] = dict(views[dom[0]], kernel=dom[0])
        roof["dp_kernels"] = views
        if n_lanes > 1:
            roof["lanes"] = {"contexts": n_lanes, "note": "the draft and the polishing call are dealt to %d contexts of this device (two host threads, two HIP streams: _capi.Api lanes); launches of the two overlap on the device, and the HIP-event bracket of a launch (`avg_launch_ms`, `kernels`, `dp_kernels`; rocprofv3's durations alike) then includes what it waited for the other lane - the fractions above are lower bounds" % n_lanes}
            if one_lane and one_lane.get(dom[0]):
                ms1 = one_lane[dom[0]]; ol = {"what": "one more step in ONE context, untimed for `value`: every kernel's time alone (what the committed counter passes and single-lane rocprof summaries match)",
                                              "kernel_ms_per_step": {k_: round(v_, 2) for k_, v_ in sorted(one_lane.items(), key=lambda kv: -kv[1])[:6]},
                                              "frac": round(alg_step.get(dom[0], n * 2.0 * L) / (ms1 / 1e3) / 8e12, 6)}
                vv = views.get(dom[0])
                if vv and vv.get("frac") and vv.get("kernel_ms"): ol["valu_issue_frac"] = round(vv["frac"] * (vv["kernel_ms"] / args.steps) / ms1, 4)
                roof["one_lane"] = ol
    # ---- host buffers in (the C-ABI takes either): ONE copy of the read set over PCIe (ngsid_reads_upload) + one pass of the hot path on it; reported beside `value`, never as it
    host_leg = None
    if not args.no_extra_step and world == 1:
        try:
            hrs_ = ReadSet(rd["seq"].cpu().numpy(), rd["qual"].cpu().numpy(), rd["off"].cpu().numpy().astype(np.uint64))
            torch.cuda.synchronize(); t_ = time.perf_counter()
            drs_ = api.upload_reads(hrs_); torch.cuda.synchronize(); up_s = time.perf_counter() - t_
            r_ = pipeline.run_hot_path(api, drs_, rd["score"], acc_rank=acc_rank, **kw); torch.cuda.synchronize(); tot_s = time.perf_counter() - t_
            nbytes = int(hrs_.seq.nbytes + hrs_.qual.nbytes + hrs_.off.nbytes)
            host_leg = {"what": "pageable host arrays -> ngsid_reads_upload (one PCIe copy) -> the same pass of the hot path", "upload_s": round(up_s, 4), "upload_GBps": round(nbytes / up_s / 1e9, 2), "bytes": nbytes,
                        "wall_s": round(tot_s, 4), "reads_per_s": round(n / tot_s, 1), "same_consensus": [c[3] for c in r_["centers"]] == [c[3] for c in res["centers"]]}
            if drs_ is not hrs_ and hasattr(drs_, "release"): drs_.release()
            del hrs_, drs_, r_
        except Exception as e:          # (never fails the line)
            host_leg = {"error": repr(e)}
    # ---- the drop-in surface (runs before the CPU baseline leg): FASTQ file in -> the reference's output files out (python -m ngspeciesid_amd ...), same reads, same flags as C3
    cli_leg = None
    if not args.no_cli and world == 1:
        import shutil, tempfile, argparse as _ap
        from ngspeciesid_amd import fastio, fastpath, cli as _cli
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        tmp = tempfile.mkdtemp(prefix="ngsid_bench_", dir=base)
        try:
            hseq = rd["seq"].cpu().numpy(); hqual = rd["qual"].cpu().numpy(); hoff = rd["off"].cpu().numpy().astype(np.uint64); hsp = rd["species"].cpu().numpy()
            hrs = ReadSet(hseq, hqual, hoff)
            perm = np.random.default_rng(3).permutation(n)                      # the file is NOT in score order
            names = fastio.Names.from_list(["r%d_sp%d" % (i, hsp[i]) for i in range(n)])
            fq = os.path.join(tmp, "reads.fastq"); fastio.write_fastq(fq, perm, names, hrs)
            in_bytes = os.path.getsize(fq)
            def _cg():          # CPU-quota throttling of the container during the leg (cgroup v2 cpu.stat / v1): the writers and the launch thread share one quota
                for f_ in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
                    try:
                        kv = dict(l.split() for l in open(f_).read().splitlines() if len(l.split()) == 2)
                        return {"nr_throttled": int(kv.get("nr_throttled", 0)), "throttled_ms": int(kv.get("throttled_usec", int(kv.get("throttled_time", 0)) // 1000)) / 1e3, "usage_s": int(kv.get("usage_usec", 0)) / 1e6}
                    except Exception:
                        pass
                return None
            def cli_run(tag, extra, t_=None):
                outd = os.path.join(tmp, "out_" + tag); os.makedirs(outd)
                cargs = _cli.build_parser().parse_args([cfg["preset"], "--fastq", fq, "--outfolder", outd, "--t", str(t_ or args.cli_t), "--consensus", "--racon", "--racon_iter", "3", "--abundance_ratio", str(AB_)] + extra)
                cargs.k, cargs.w = K_, W_
                time.sleep(1.0)         # (outside the timed leg: the leg before released its gigabytes of arrays half a second after its writers were joined - fastio.NativeJobs - which holds the interpreter lock)
                cg0 = _cg()
                api.__dict__["call_s"] = {}; api.lib.ngsid_profile_enable(api.ctx, C.c_int32(1))
                tcl = time.perf_counter(); r = fastpath.main(cargs, api=api); dcl = time.perf_counter() - tcl
                cg1 = _cg()
                pbuf = C.create_string_buffer(1 << 16); api.lib.ngsid_profile_read(api.ctx, pbuf, C.c_uint64(len(pbuf))); api.lib.ngsid_profile_enable(api.ctx, C.c_int32(0))
                lib_s = {l.split()[0][5:]: float(l.split()[2]) / 1e3 for l in pbuf.value.decode().splitlines() if l.startswith("host_")}
                bind = {nm: {"caller_s": round(api.call_s.get(nm, 0.0), 3), "library_s": round(v, 3), "difference_s": round(api.call_s.get(nm, 0.0) - v, 3)} for nm, v in lib_s.items()}
                out_bytes = sum(os.path.getsize(os.path.join(rt, f)) for rt, _, fs in os.walk(outd) for f in fs)
                paf_bytes = sum(os.path.getsize(os.path.join(rt, f)) for rt, _, fs in os.walk(outd) for f in fs if f.endswith(".paf"))
                got = sorted(m[2] for m in r["centers"])
                shutil.rmtree(outd, ignore_errors=True)
                return {"reads_per_s": round(n / dcl, 1), "wall_s": round(dcl, 3), "stage_s": {k_: (round(v, 3) if isinstance(v, float) else ({a_: round(b_, 3) for a_, b_ in v.items()} if isinstance(v, dict) else v)) for k_, v in r["timings"].items()}, "output_bytes": out_bytes, "paf_bytes": paf_bytes,
                        "consensus_equals_amplicons": got == sorted(truths), "binding_overhead_s": bind,
                        "cgroup_cpu_during_the_leg": None if not (cg0 and cg1) else {"throttled_periods": cg1["nr_throttled"] - cg0["nr_throttled"], "throttled_ms": round(cg1["throttled_ms"] - cg0["throttled_ms"], 1), "cpu_seconds_used": round(cg1["usage_s"] - cg0["usage_s"], 2)}}
            # three legs on the same file: the CLI as a user runs it (polishing of a cluster stops once an iteration returns its input: the library default), the same with EVERY
            # iteration (--polish_all_iterations: equal work to `value`, which runs all three - VERDICT r5 item 9), and without the PAF files (--skip_paf: what they cost)
            leg_stop = cli_run("stop", []); leg_all = cli_run("all", ["--polish_all_iterations"]); leg_nopaf = cli_run("nopaf", ["--polish_all_iterations", "--skip_paf"])
            leg_t8 = cli_run("t8", [], t_=8) if args.cli_t == 1 else None          # the reference's DEFAULT --t 8: eight score-ordered batches + merge rounds (another clustering than --t 1, the reference's own for 8 cores)
            cli_leg = dict(leg_all)
            cli_leg.update({"ratio_to_hot_path": round(leg_all["reads_per_s"] / reads_per_s, 3), "t": args.cli_t, "input_fastq_bytes": in_bytes,
                            "files_on": "tmpfs (/dev/shm)" if base else "disk (tmp dir)",
                            "what": "python -m ngspeciesid_amd %s --fastq reads.fastq --outfolder out --t %d --consensus --racon --racon_iter 3 --abundance_ratio %s --polish_all_iterations: FASTQ parse, score, sort, sorted.fastq, "
                                    "clustering, final_clusters.tsv / final_cluster_origins.tsv, draft consensus, rc merge, consensus_reference_*.fasta, reads_to_consensus_*.fastq, ALL 3 polishing iterations (as `value`), "
                                    "racon_cl_id_*/{consensus.fasta, racon_polished_it_*.fasta, read_alignments_it_*.paf}" % (cfg["preset"], args.cli_t, AB_),
                            "with_stable_stop": {"reads_per_s": leg_stop["reads_per_s"], "wall_s": leg_stop["wall_s"], "stage_s": leg_stop["stage_s"], "consensus_equals_amplicons": leg_stop["consensus_equals_amplicons"],
                                                 "ratio_to_hot_path_with_stable_stop": round(leg_stop["reads_per_s"] / (n_total / dt_stop), 3) if dt_stop > 0 else None,
                                                 "what": "the CLI's default (no --polish_all_iterations) against the hot path with the same early stop (config.with_stable_stop)"},
                            "without_paf": {"reads_per_s": leg_nopaf["reads_per_s"], "wall_s": leg_nopaf["wall_s"], "what": "--polish_all_iterations --skip_paf: the cost of writing minimap2's PAF of every iteration is the difference to this leg"}})
            if leg_t8 is not None:
                cli_leg["with_the_references_default_t8"] = {"reads_per_s": leg_t8["reads_per_s"], "wall_s": leg_t8["wall_s"], "cluster_s": leg_t8["stage_s"].get("cluster"), "consensus_equals_amplicons": leg_t8["consensus_equals_amplicons"],
                                                             "what": "--t 8 (the reference's default) with the early stop: the 8-batch schedule of parallelize.py on one GPU - 8 + 4 + 2 + 1 clustering calls, the batch of the worst reads alone makes ~800 representatives; --t 1 is the fast setting here (INTEGRATION.md)"}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    # ---- CPU baseline: the oracle (a scalar port of the same algorithms) on a bounded sample of the same workload: one core, and all host cores
    #      with one batch per core like the reference's `--t N` worker processes (merge rounds not included: they are O(representatives))
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import subprocess, tempfile, shutil
        from oracle_lib import load_oracle
        from ngspeciesid_amd import parallelize
        orc = load_oracle()
        cores = os.cpu_count() or 1
        usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores          # what this process may actually run on ...
        try:                                                                                           # ... and the container's CPU quota (cgroup v2 / v1)
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            if q[0] != "max": usable = min(usable, max(1, int(float(q[0]) / float(q[1]))))
        except Exception:
            try:
                qq = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); pp = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if qq > 0: usable = min(usable, max(1, qq // pp))
            except Exception:
                pass
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:
            model = "unknown"
        per_core = max(200, args.cpu_sample)
        use = max(1, min(usable, args.cpu_cores if args.cpu_cores > 0 else usable))
        ns = min(n, per_core * use)
        idx = np.linspace(0, n - 1, ns).astype(np.int64)          # evenly strided subset: keeps the score order and the species mix
        seq = rd["seq"].cpu().numpy(); qual = rd["qual"].cpu().numpy(); off = rd["off"].cpu().numpy()
        from ngspeciesid_amd.hostutil import subset_reads
        srs = subset_reads(ReadSet(seq, qual, off.astype(np.uint64)), idx)
        kwc = {k_: v for k_, v in kw.items() if k_ != "p_shared"}
        # (a) one core, one batch of `per_core` reads
        one = subset_reads(srs, np.arange(min(per_core, ns)))
        tc = time.perf_counter()
        pipeline.run_hot_path(orc, one, rd["score"][idx][:one.n], acc_rank=acc_rank[idx][:one.n], **kw)
        dt1 = time.perf_counter() - tc
        # (b) all cores: one worker process per batch of the reference's total_nt partition
        tmpd = tempfile.mkdtemp(prefix="ngsid_cpu_")
        allc = None
        try:
            batches = [bb for bb in parallelize.batch_list_total_nt(np.diff(srs.off.astype(np.int64)), use) if bb[1] > bb[0]]
            for i, (a_, b_) in enumerate(batches):       # one small sample file per worker (a worker must not page in the whole sample)
                bsub = subset_reads(srs, np.arange(a_, b_))
                np.savez(os.path.join(tmpd, "s%d.npz" % i), seq=bsub.seq, qual=bsub.qual, off=bsub.off, score=rd["score"][idx][a_:b_], acc_rank=acc_rank[idx][a_:b_], p_shared=ptab, kw=json.dumps(kwc))
            tc = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), os.path.join(tmpd, "s%d.npz" % i), "0", str(b_ - a_), os.path.join(tmpd, "o%d.json" % i)],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for i, (a_, b_) in enumerate(batches)]
            deadline = time.perf_counter() + 120.0           # bounded leg: workers still running after two minutes are stopped (by PID) and the leg reports nothing
            for p_ in procs:
                try: p_.wait(timeout=max(0.1, deadline - time.perf_counter()))
                except subprocess.TimeoutExpired: p_.kill(); p_.wait()
            dta = time.perf_counter() - tc
            done = [json.load(open(os.path.join(tmpd, "o%d.json" % i))) for i in range(len(batches)) if os.path.exists(os.path.join(tmpd, "o%d.json" % i))]
            if len(done) == len(batches):
                allc = {"value": round(ns / dta, 1), "cores": len(batches), "wall_s": round(dta, 1), "slowest_worker_s": round(max(d_["seconds"] for d_ in done), 1)}
        finally:
            shutil.rmtree(tmpd, ignore_errors=True)
        cpu = {"value": allc["value"] if allc else round(one.n / dt1, 2), "unit": "reads/s", "cores": allc["cores"] if allc else 1, "kind": "port",
               "reference_python_one_core": {"reads_to_clusters_reads_per_s": 294, "get_sorted_fastq_reads_per_s": 6200, "box": "build container, Intel Xeon @ 2.10 GHz, one core, 4 000 reads of this profile (oracle/time_reference.py; BASELINE.md section 5)",
                                             "note": "the reference's own Python for the CLUSTERING half only (its aligner behind a parasail-shaped shim = the oracle's scalar C aligner); spoa / racon / minimap2 are not in the image, the reference cannot run on the GPU box - a recorded constant, not measured by this run"},
               "cpu_model": model, "host_cores": cores, "usable_cores": usable,
               "one_core": {"value": round(one.n / dt1, 2), "reads": int(one.n), "seconds": round(dt1, 1)}, "all_cores": allc,
               "sample": "%d reads strided from the same batch (same params, tile_depth %d): %d per worker process, one process per usable core (affinity mask / cgroup quota; start-up of the interpreters included in the wall time) running oracle/libngsid_oracle.so "
                         "(scalar C port of this build's algorithms; the reference's own tools - parasail, spoa, racon - are SIMD codes and are not in the image, see BASELINE.md)"
                         % (ns, args.tile_depth, per_core)}
    out = {"metric": "reads/sec end-to-end (cluster + spoa consensus + racon x3), %d bp %s" % (args.length, "CCS" if cfg["preset"] == "--isoseq" else "ONT"), "value": round(reads_per_s, 1), "unit": "reads/s",
           "n_gpus": world, "rccl_ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
           "scaling": args.scaling if world > 1 or force_dist else "weak", "vs_baseline": None, "dtype": "u8 / int16 / int32 DP, 64-bit bit-vectors (f64 thresholds)", "data": "synthetic" + (" (counter-based generator, no torch.Generator)" if RNG_ == "hash" else ""),
           "config": {"workload": (args.config.upper() + ": %d synthetic %d bp %s-profile reads " + ("in total, one `--t N` batch per GPU" if (args.scaling == "strong" and (world > 1 or force_dist)) else "per GPU") + " (mu=%.0f), %d species @15%% divergence%s, k=%d w=%d, cluster + spoa-style POA + racon-style polish x3, abundance_ratio %s, POA tile depth %d band %s")
                      % (args.reads, args.length, "CCS" if args.mu >= 25 else "ONT", args.mu, args.species, (" with geometric abundance %.1f^i" % cfg["geometric"]) if cfg["geometric"] else "", K_, W_, AB_, args.tile_depth, ("%d" % args.band) if args.band else "%d (library default, widened per tile by the band-edge check)" % (64 if args.length <= 3000 else 128)),
                      "parallelism": ("1 GPU" if world == 1 else "%d shards (one per GPU), RCCL all-gather of representatives + partial consensuses" % world),
                      "poa_single_below": pipeline.SINGLE_BELOW, "reads_clustered_per_gpu": n, "f_aln": round(f_aln, 4), "stage_s_per_step": {k_: round(v / args.steps, 4) for k_, v in T.items()}, "stage_s_per_step_max_over_ranks": None if stage_max is None else {k_: round(v / args.steps, 4) for k_, v in stage_max.items()},
                      "kernel_ms_per_step": {k_: round(v[1] / args.steps, 2) for k_, v in kern.items()}, "kernel_ms_per_step_one_lane": None if not one_lane else {k_: round(v_, 2) for k_, v_ in one_lane.items()}, "lanes": n_lanes, "library_host_ms_per_step": {k_: round(v[1] / args.steps, 2) for k_, v in host_lines.items()}, "poa_tiles_redone_with_wider_band_per_step": round(redo_tiles / args.steps, 1),
                      "hbm_gb": {"peak_in_timed_steps": round(hbm_peak / 1e9, 2), "held_after": round(hbm_live / 1e9, 2), "what": "bytes handed out by the library's device allocator, whole process (read set included)", "context_scratch_by_purpose": mem_parts},
                      "check": {"cluster_purity": round(purity, 5), "centers": len(big), "consensus_edit_distance_vs_truth": ed, "membership_equals_reference_t_n": membership_ok, "sharded_consensus_equals_single_process": single_same}},
           "roofline": roof, "cpu_baseline": cpu}
    if dist is not None:
        out["config"]["ranks"] = {"world_size": world, "ranks_in_the_all_reduce": ranks_seen, "distinct_gpus": gpus_seen, "backend": backend + (" (= RCCL)" if backend == "nccl" else " (dev aid: collectives on the host, ranks may share a GPU)")}
    if world > 1:
        out["config"]["baseline_config_of_this_line"] = ("%s of BASELINE.json, %s" % (args.config.upper(), "ONE global set split into the reference's `--t %d` batches (strong scaling)" % world if args.scaling == "strong" else
                                                         "its per-GPU shape repeated on every GPU with an independent read set per rank (weak scaling%s)" % ("; C3 is BASELINE's 1-GPU configuration" if args.config == "c3" else "")))
        out["config"]["eight_gpu_configurations"] = "BASELINE.json quotes C4 (10 M x 750 bp, 50 species) and C5 (2 M x 2 kb CCS, 20 species) on 8 GPUs as ONE global set: python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 bench.py --gpus 8 --config c4 --scaling strong --check-membership   (likewise --config c5)"
    if cli_leg is not None: out["config"]["cli"] = cli_leg
    if host_leg is not None: out["config"]["host_buffers_in"] = host_leg
    if res_stop is not None: out["config"]["with_stable_stop"] = {"reads_per_s": round(n_total / dt_stop, 1), "ms_per_step": round(dt_stop * 1e3, 2), "same_result": stop_same,
                                          "note": "library default stop_when_stable=1 (not used for `value`): a cluster whose polished sequence equals its backbone is not polished again"}
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        try: dist.destroy_process_group()
        except Exception: pass


if __name__ == "__main__":
    main()
