"""Test helper: drives the reference-shaped FUNCTIONS of the package - get_sorted_fastq_for_cluster.main, cluster.reads_to_clusters, parallelize.parallel_clustering,
consensus.form_draft_consensus / detect_reverse_complements / polish_sequences (the Python surface a caller of the reference's modules binds to) - and leaves the
files of the CLI contract in args.outfolder, so that tests can compare them with what the array path of the CLI (fastpath.py, the product's only `main`) writes.
Round 6: this replaces `cli.main_reference_shaped` (a second `main` inside the product that followed the reference's script line by line and could not trim primers)."""
import os, random, shutil, tempfile
from ngspeciesid_amd import get_sorted_fastq_for_cluster, parallelize, cluster, consensus, help_functions
from ngspeciesid_amd.ptable import p_emp_probs_dict


def _strip(acc):
    return acc.rsplit("_", 1)[0]


def _score_of(acc):
    return float(acc.rsplit("_", 1)[1])


def run(args):
    args.outfile = os.path.join(args.outfolder, "sorted.fastq")
    path = get_sorted_fastq_for_cluster.main(args)
    with open(path) as fh:
        reads = [(i, 0, acc, s, q, _score_of(acc)) for i, (acc, (s, q)) in enumerate(help_functions.readfq(fh))]
    lo, hi = args.target_length - args.target_deviation, args.target_length + args.target_deviation
    if args.target_length > 0 and args.target_deviation > 0:
        reads = [r for r in reads if lo <= len(r[3]) <= hi]
    if args.top_reads:
        reads = reads[:args.sample_size]
    elif 0 < args.sample_size < len(reads):
        reads = [reads[i] for i in sorted(random.sample(range(len(reads)), args.sample_size))]
    cutoff = int(args.abundance_ratio * len(reads))
    table = p_emp_probs_dict(args.k, args.w)
    if args.nr_cores > 1:
        clusters, reps = parallelize.parallel_clustering(reads, table, args)
    else:
        clusters = {r[0]: [r[2]] for r in reads}; reps = {r[0]: r for r in reads}
        clusters, reps, _, _ = next(iter(cluster.reads_to_clusters(clusters, reps, reads, table, {}, 1, args).values()))
    ranked = sorted(clusters, key=lambda c: (len(clusters[c]), reps[c][5]), reverse=True)
    with open(os.path.join(args.outfolder, "final_clusters.tsv"), "w") as tsv, open(os.path.join(args.outfolder, "final_cluster_origins.tsv"), "w") as org:
        for out_id, c in enumerate(ranked):
            t = reps[c]
            org.write("\t".join([str(out_id), _strip(t[2]), t[3], t[4], str(t[5]), str(t[6]) if len(t) == 8 else ""]) + "\n")
            tsv.writelines("%d\t%s\n" % (out_id, _strip(a)) for a in sorted(clusters[c], key=_score_of, reverse=True))
    if args.consensus:
        work = tempfile.mkdtemp()
        try:
            centers = consensus.form_draft_consensus(clusters, reps, path, work, cutoff, args)
            consensus.polish_sequences(consensus.detect_reverse_complements(centers, args.rc_identity_threshold), args)
        finally:
            shutil.rmtree(work)
    return clusters, reps
