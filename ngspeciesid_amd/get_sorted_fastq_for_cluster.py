"""Read scoring + filtering + stable sort (reference modules/get_sorted_fastq_for_cluster.py:124-191) with the scores from ngsid_score_reads."""
from __future__ import annotations
import logging
import os
import numpy as np
from . import runtime
from ._capi import ReadSet
from .help_functions import readfq


def main(args, api=None):
    api = api or runtime.get_api()
    if os.path.isfile(args.outfile) and getattr(args, "use_old_sorted_file", False):
        logging.warning("Using already existing sorted file in specified directory, in not intended, specify different outfolder or delete the current file.")
        return args.outfile
    with open(args.fastq) as f:
        recs = [(acc, seq, qual) for acc, (seq, qual) in readfq(f)]
    rs = ReadSet.from_strings([r[1] for r in recs], [r[2] for r in recs])
    score, err, keep = api.score_reads(rs, args.k, args.quality_threshold)
    idx = np.nonzero(keep)[0]
    order = idx[np.argsort(-score[idx], kind="stable")]                      # read_array.sort(key=score, reverse=True) is stable
    with open(args.outfile, "w") as out:
        for i in order:
            acc, seq, qual = recs[i]
            out.write("@{0}\n{1}\n+\n{2}\n".format(acc + "_{0}".format(float(score[i])), seq, qual))
    logging.debug(f"{len(order)} reads passed quality critera (avg phred Q val over {args.quality_threshold} and length > 2*k) and will be clustered.")
    er = np.sort(err[idx])
    with open(os.path.join(args.outfolder, "logfile.txt"), "w") as lf:
        if len(er):
            lf.write("Lowest read error rate:{0}\n".format(float(er[0])))
            lf.write("Highest read error rate:{0}\n".format(float(er[-1])))
            lf.write("Median read error rate:{0}\n".format(float(er[int(len(er) / 2)])))
            lf.write("Mean read error rate:{0}\n".format(float(sum(er.tolist()) / len(er))))
        lf.write("\n")
    return args.outfile
