"""Reference-shaped consensus / polishing interface (modules/consensus.py).

run_spoa / run_racon keep the reference's signatures and file contracts (consensus.py:83-92,107-126) but call the HIP
library instead of spawning `spoa`, `minimap2` and `racon`; form_draft_consensus, detect_reverse_complements and
polish_sequences mirror consensus.py:249-278,148-183,186-246 on top of them.  medaka is out of scope.
"""
from __future__ import annotations
import glob
import logging
import os
import shutil
import numpy as np
from . import runtime
from . import pipeline
from ._capi import ReadSet, poa_params, polish_params, POA_LOCAL
from .help_functions import readfq, mkdir_p

DEFAULT_TILE_DEPTH = pipeline.TILE_DEPTH
DEFAULT_BAND = 0          # library default: 64 columns for reads up to 1 024 bases, 128 beyond; widened per tile by the band-edge check


def reverse_complement(string):
    rev_nuc = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A', 'a': 't', 'c': 'g', 'g': 'c', 't': 'a', 'N': 'N', 'X': 'X', 'n': 'n', 'Y': 'R', 'R': 'Y', 'K': 'M', 'M': 'K',
               'S': 'S', 'W': 'W', 'B': 'V', 'V': 'B', 'H': 'D', 'D': 'H', 'y': 'r', 'r': 'y', 'k': 'm', 'm': 'k', 's': 's', 'w': 'w', 'b': 'v', 'v': 'b', 'h': 'd', 'd': 'h'}
    return ''.join([rev_nuc[nucl] for nucl in reversed(string)])


def _read_fastx(path):
    accs, seqs, quals = [], [], []
    with open(path) as f:
        for acc, (seq, qual) in readfq(f):
            accs.append(acc); seqs.append(seq); quals.append(qual)
    return accs, seqs, quals


def run_spoa(reads, spoa_out_file, spoa_path, api=None, tile_depth=DEFAULT_TILE_DEPTH, band=DEFAULT_BAND, single_below=None):
    """`spoa reads -l 0 -r 0 -g -2` (consensus.py:87): reads a FASTQ/FASTA file, writes the 2-line FASTA spoa prints, returns line 2."""
    api = api or runtime.get_api()
    accs, seqs, quals = _read_fastx(reads)
    rs = ReadSet.from_strings(seqs, quals if all(q is not None for q in quals) and quals else None)
    node_cap = 0 if max((len(s) for s in seqs), default=0) <= 1000 else 22
    consensus = api.poa_consensus(rs, [0, len(seqs)], poa_params(mode=POA_LOCAL, match=5, mismatch=-4, gap=-2, tile_depth=tile_depth, band=band, node_cap=node_cap, trim=pipeline.DRAFT_TRIM,
                                                                 single_below=pipeline.SINGLE_BELOW if single_below is None else single_below))[0]
    with open(spoa_out_file, "w") as f:
        f.write(">Consensus LN:i:{0}\n{1}\n".format(len(consensus), consensus))
    return consensus


def run_racon(reads_to_center, center_file, outfolder, cores, racon_iter, api=None, tile_depth=DEFAULT_TILE_DEPTH, band=DEFAULT_BAND, k=13, w=20, trim=2, single_below=None):
    """(minimap2 -x map-ont -> racon) x racon_iter (consensus.py:107-126): writes outfolder/racon_polished_it_{i}.fasta for every iteration and consensus.fasta."""
    api = api or runtime.get_api()
    accs, seqs, quals = _read_fastx(reads_to_center)
    caccs, cseqs, _ = _read_fastx(center_file)
    rs = ReadSet.from_strings(seqs, quals if quals and all(q is not None for q in quals) else None)
    node_cap = 0 if max((len(s) for s in seqs), default=0) <= 1000 else 22
    prm = polish_params(iters=racon_iter, k=k, w=w, tile_depth=tile_depth, band=band, node_cap=node_cap, trim=trim, single_below=pipeline.SINGLE_BELOW if single_below is None else single_below)
    with open(os.path.join(outfolder, "stdout.txt"), "w") as f:
        f.write("")
    name = caccs[0].split()[0]
    if racon_iter >= 1:
        its, used, aln = api.polish_trace(ReadSet.from_strings([cseqs[0]]), rs, [0, len(seqs)], prm, aln=True)
        write_racon_iteration_files(outfolder, name, [x[0] for x in its], [int(u[0]) for u in used])
        from . import fastio                            # read_alignments_it_{i}.paf: the alignments iteration i polished with (consensus.py:112-121)
        names = fastio.Names.from_list([a.split()[0] for a in accs]); ids = np.arange(len(seqs))
        for i in range(racon_iter):
            fastio.write_paf(os.path.join(outfolder, "read_alignments_it_{0}.paf".format(i)), ids, names, rs.off, aln[i], name, len(cseqs[0] if i == 0 else its[i - 1][0]))
    else:                                                       # no iteration: consensus.fasta is the centre file itself (consensus.py:124)
        shutil.copyfile(center_file, os.path.join(outfolder, "consensus.fasta"))


def write_racon_iteration_files(outfolder, name, seqs, used):
    """the files run_racon leaves in racon_cl_id_X/ (consensus.py:110-124): racon_polished_it_{i}.fasta after every iteration (racon's header tags),
    the (empty) stderr files of the two tools, consensus.fasta = the last iteration.  read_alignments_it_{i}.paf is written by the caller from the records of
    ngsid_polish_trace_aln (fastio.write_paf)."""
    last = None
    for i, (s, u) in enumerate(zip(seqs, used)):
        for fn in ("mm2_stderr_it_{0}.txt", "racon_stderr_it_{0}.txt"):
            open(os.path.join(outfolder, fn.format(i)), "w").close()
        last = os.path.join(outfolder, "racon_polished_it_{0}.fasta".format(i))
        with open(last, "w") as f:
            f.write(">{0} LN:i:{1} RC:i:{2} XC:f:1.000000\n{3}\n".format(name, len(s), u, s))
    if last is not None:
        shutil.copyfile(last, os.path.join(outfolder, "consensus.fasta"))


def highest_aln_identity(seq, seq2, api=None):
    """max identity over forward / reverse-complement semi-global alignments (consensus.py:129-145)."""
    api = api or runtime.get_api()
    q = ReadSet.from_strings([seq]); t = ReadSet.from_strings([seq2, reverse_complement(seq2)])
    score, ncols, nmatch, _ = api.sg_align_batch(q, t, [0, 0], [0, 1], 3, 1, 2, -2, 13, None)
    return max(nmatch[0] / float(ncols[0]), nmatch[1] / float(ncols[1]))


def form_draft_consensus(clusters, representatives, sorted_reads_fastq_file, work_dir, abundance_cutoff, args, api=None):
    centers, singletons, discarded = [], 0, []
    with open(sorted_reads_fastq_file) as f:
        reads = {acc: (seq, qual) for acc, (seq, qual) in readfq(f)}
    for c_id, all_read_acc in sorted(clusters.items(), key=lambda x: (len(x[1]), representatives[x[0]][5]), reverse=True):
        n = len(all_read_acc)
        if n >= abundance_cutoff:
            reads_path_name = os.path.join(work_dir, "reads_c_id_{0}.fq".format(c_id))
            with open(reads_path_name, "w") as rf:
                for i, acc in enumerate(all_read_acc):
                    if args.max_seqs_for_consensus >= 0 and i >= args.max_seqs_for_consensus:
                        break
                    seq, qual = reads[acc]
                    rf.write("@{0}\n{1}\n{2}\n{3}\n".format(acc, seq, "+", qual))
            center = run_spoa(reads_path_name, os.path.join(work_dir, "spoa_tmp.fa"), "spoa", api=api, tile_depth=getattr(args, "poa_tile_depth", DEFAULT_TILE_DEPTH), band=getattr(args, "poa_band", DEFAULT_BAND), single_below=getattr(args, "poa_single_below", None))
            centers.append([n, c_id, center, reads_path_name])
        elif n == 1:
            singletons += 1
        elif n > 1:
            discarded.append(n)
    logging.debug(f"{singletons} singletons were discarded")
    logging.debug(f"{len(discarded)} clusters were discarded due to not passing the abundance_cutoff: a total of {sum(discarded)} reads were discarded. "
                  f"Highest abundance among them: {max(discarded or [0])} reads.")
    return centers


def detect_reverse_complements(centers, rc_identity_threshold, api=None):
    api = api or runtime.get_api()
    cs = [[c[0], c[1], c[2], c[3] if isinstance(c[3], list) else [c[3]]] for c in centers]
    merged = pipeline.detect_reverse_complements(api, cs, rc_identity_threshold)
    logging.debug(f"{len(merged)} consensus formed.")
    return merged


def polish_sequences(centers, args, api=None):
    if getattr(args, "medaka", False):
        raise NotImplementedError("--medaka (neural polisher) is out of scope of the MI355X hot path; use --racon")
    for folder in glob.glob(os.path.join(args.outfolder, "racon_cl_id_*")):
        shutil.rmtree(folder)
    for file in glob.glob(os.path.join(args.outfolder, "consensus_reference_*")):
        os.remove(file)
    for i, (nr_reads_in_cluster, c_id, center, all_reads) in enumerate(centers):
        spoa_center_file = os.path.join(args.outfolder, "consensus_reference_{0}.fasta".format(c_id))
        with open(spoa_center_file, "w") as f:
            f.write(">{0}\n{1}\n".format("consensus_cl_id_{0}_total_supporting_reads_{1}".format(c_id, nr_reads_in_cluster), center))
        all_reads_file = os.path.join(args.outfolder, "reads_to_consensus_{0}.fastq".format(c_id))
        nr_reads_used = 0
        with open(all_reads_file, "w") as f:
            for fasta_file in all_reads:
                with open(fasta_file) as rf:
                    reads = {acc: (seq, qual) for acc, (seq, qual) in readfq(rf)}
                for acc, (seq, qual) in reads.items():
                    f.write("@{0}\n{1}\n{2}\n{3}\n".format(acc.split()[0], seq, "+", qual)); nr_reads_used += 1
        if getattr(args, "racon", False):
            logging.debug("running racon on spoa reference {0} using {1} reads for polishing.".format(c_id, nr_reads_used))
            folder = os.path.join(args.outfolder, "racon_cl_id_{0}".format(c_id))
            mkdir_p(folder)
            run_racon(all_reads_file, spoa_center_file, folder, "1", args.racon_iter, api=api, tile_depth=(getattr(args, "poa_tile_depth", 0) if getattr(args, "poa_tile_depth", 0) > 0 else DEFAULT_TILE_DEPTH), band=getattr(args, "poa_band", DEFAULT_BAND), k=args.k, w=args.w, single_below=getattr(args, "poa_single_below", None))
            with open(os.path.join(folder, "consensus.fasta")) as cf:
                centers[i][2] = cf.readlines()[1].strip()
    return centers
