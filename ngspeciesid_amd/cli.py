"""Command line of the reference (NGSpeciesID:187-287) over the MI355X hot path:  python -m ngspeciesid_amd --ont --fastq X --outfolder O --consensus --racon

Same flags and defaults, same output files (sorted.fastq, logfile.txt, final_clusters.tsv, final_cluster_origins.tsv,
consensus_reference_*.fasta, reads_to_consensus_*.fastq, racon_cl_id_*/consensus.fasta), including --primer_file /
--remove_universal_tails (barcode_trimmer.py).  --medaka is outside the hot path and is refused.
"""
from __future__ import annotations
import argparse, logging, os, sys


def main(args, api=None):
    """The CLI's work = the array path (fastpath.py).  The dict / file functions with the reference's Python signatures (cluster.reads_to_clusters, parallelize.parallel_clustering,
    consensus.run_spoa / run_racon / form_draft_consensus / polish_sequences) stay importable for callers of the reference's modules; tests/dict_layer.py drives them and compares
    the files they leave with the ones written here."""
    from . import fastpath
    return fastpath.main(args, api=api)


def write_fastq(args):
    """the `write_fastq` sub-command (NGSpeciesID:161-182, :238-245): one <cluster id>.fastq per cluster of final_clusters.tsv with at least --N reads.  The reference's
    semantics: the id and the accession are the first two white-space separated fields of a line, the reads are looked up by their WHOLE header line (an accession the FASTQ does
    not hold under that name is a KeyError, as there)."""
    from .help_functions import readfq, mkdir_p
    members = {}
    with open(args.clusters) as fh:
        for line in fh:
            f = line.split()
            if len(f) >= 2:
                members.setdefault(f[0], []).append(f[1])
    mkdir_p(args.outfolder)
    with open(args.fastq) as fh:
        record = {name: sq for name, sq in readfq(fh)}
    for cl_id, accs in members.items():
        if len(accs) < args.N:
            continue
        with open(os.path.join(args.outfolder, str(cl_id) + ".fastq"), "w") as out:
            for acc in accs:
                seq, qual = record[acc]
                out.write("@{0}\n{1}\n+\n{2}\n".format(acc, seq, qual))


def build_parser():
    p = argparse.ArgumentParser(description="Reference-free clustering and consensus forming of targeted ONT or PacBio reads (MI355X hot path)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('--version', action='version', version='%(prog)s 0.3.1-mi355x')
    p.add_argument('--debug', action='store_true')
    rf = p.add_mutually_exclusive_group(required=True)
    rf.add_argument('--fastq', type=str)
    rf.add_argument('--use_old_sorted_file', action='store_true')
    p.add_argument('--t', dest="nr_cores", type=int, default=8, help='Number of score-ordered batches (the reference\'s cores); the cluster membership depends on it exactly like in the reference')
    p.add_argument('--d', dest="print_output", type=int, default=10000)
    p.add_argument('--q', dest="quality_threshold", type=float, default=7.0)
    p.add_argument('--ont', action="store_true"); p.add_argument('--isoseq', action="store_true")
    p.add_argument('--consensus', action="store_true")
    p.add_argument('--abundance_ratio', type=float, default=0.1)
    p.add_argument('--rc_identity_threshold', type=float, default=0.9)
    p.add_argument('--max_seqs_for_consensus', type=int, default=-1)
    g = p.add_mutually_exclusive_group()
    g.add_argument('--medaka', action="store_true"); g.add_argument('--racon', action="store_true")
    p.add_argument('--medaka_model', type=str, default=""); p.add_argument('--medaka_fastq', action="store_true")
    p.add_argument('--racon_iter', type=int, default=2)
    g2 = p.add_mutually_exclusive_group()
    g2.add_argument('--remove_universal_tails', action="store_true"); g2.add_argument('--primer_file', type=str, default="")
    p.add_argument('--primer_max_ed', type=int, default=2); p.add_argument('--trim_window', type=int, default=150)
    p.add_argument('--m', dest="target_length", type=int, default=0); p.add_argument('--s', dest="target_deviation", type=int, default=0)
    p.add_argument('--sample_size', type=int, default=0); p.add_argument('--top_reads', action='store_true')
    p.add_argument('--k', type=int, default=13); p.add_argument('--w', type=int, default=20)
    p.add_argument('--min_shared', type=int, default=5); p.add_argument('--mapped_threshold', type=float, default=0.7)
    p.add_argument('--aligned_threshold', type=float, default=0.4); p.add_argument('--symmetric_map_align_thresholds', action='store_true')
    p.add_argument('--batch_type', type=str, default='total_nt'); p.add_argument('--min_fraction', type=float, default=0.8)
    p.add_argument('--min_prob_no_hits', type=float, default=0.1); p.add_argument('--outfolder', type=str, default=None)
    # extensions of this build (not in the reference): shape of the consensus engine.  Defaults = the measured configuration.
    p.add_argument('--poa_tile_depth', type=int, default=4, help='reads per exact-order POA tile (default 4: depth-tiled hierarchy; a positive value also applies to the polishing windows). 0 = one graph per cluster in read order, i.e. spoa\'s order; a graph holds at most 65 520 nodes, so use it with --max_seqs_for_consensus (a few hundred reads); larger clusters are split by the capacity rule')
    p.add_argument('--strand_aware', action='store_true', help='extension: clusters whose representatives are reverse complements of each other (decided by the clustering criteria themselves) are joined BEFORE the consensus stage and their reads are oriented: one cluster per amplicon in final_clusters.tsv; off = the reference (two clusters per amplicon on mixed-strand data, joined after the drafts)')
    p.add_argument('--polish_all_iterations', action='store_true', help='extension: run every --racon_iter iteration even when an iteration returned its input unchanged (the default stops polishing such a cluster: same result, less time)')
    p.add_argument('--poa_single_below', type=int, default=None, help='extension: clusters / polishing windows with fewer sequences than this are aligned as ONE graph in read order (spoa\'s and racon\'s own order) instead of being depth-tiled; 0 = tile everything; default: the library\'s measured threshold (pipeline.SINGLE_BELOW)')
    p.add_argument('--skip_paf', action='store_true', help='extension: do not write racon_cl_id_*/read_alignments_it_{i}.paf (the reference leaves minimap2\'s PAF of every polishing iteration there; default: written)')
    p.set_defaults(which='main')
    sub = p.add_subparsers(help='sub-command help')
    wf = sub.add_parser('write_fastq', help='write the reads of every cluster of final_clusters.tsv to <outfolder>/<cluster id>.fastq (NGSpeciesID:238-245)')
    wf.add_argument('--clusters', type=str, help='final_clusters.tsv of a run')
    wf.add_argument('--fastq', type=str, help='Input fastq file')
    wf.add_argument('--outfolder', type=str, help='Output folder')
    wf.add_argument('--N', type=int, default=0, help='Write out clusters with more or equal than N reads')
    wf.set_defaults(which='write_fastq')
    p.add_argument('--poa_band', type=int, default=0, help='band of the POA alignments in columns (0 = library default: 64 for reads up to 3 kb, else 128; a tile whose path touches the band edge is redone at twice the band)')
    return p


def cli(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.debug else logging.INFO, format='%(message)s')
    if getattr(args, "which", "main") == 'write_fastq':          # NGSpeciesID:255-258
        write_fastq(args)
        logging.info("Wrote clusters to separate fastq files.")
        sys.exit(0)
    if args.ont and args.isoseq:
        logging.error("Arguments mutually exclusive, specify either --isoseq or --ont. "); sys.exit()
    elif args.isoseq:
        args.k, args.w = 15, 50
    elif args.ont:
        args.k, args.w = 13, 20
    if args.medaka:
        logging.error("--medaka (neural polisher) is outside the accelerated hot path (see DESIGN.md); use --racon."); sys.exit(1)
    if args.k > 32 or args.k < 1:
        logging.error('k = %d is outside what the minimizer encoder of this build handles (1..32; the shared-minimizer table has rows for k = 10..30).' % args.k); sys.exit(1)
    if 100 < args.w or args.w < args.k:
        logging.error('Please specify a window of size larger or equal to k, and smaller than 100.'); sys.exit(1)
    if args.poa_single_below is None:
        from . import pipeline
        args.poa_single_below = pipeline.SINGLE_BELOW
    if args.outfolder and not os.path.exists(args.outfolder):
        os.makedirs(args.outfolder)
    main(args)


if __name__ == "__main__":
    cli()
