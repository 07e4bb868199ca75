"""Regression pin of the consensus half on the reference's own reads (tests/golden/sample_h1_consensus_oracle.json).

NOT a reference vector: spoa / racon / minimap2 are absent, so nothing the reference holds pins this half.  This file records what
THIS build's oracle returns for `--ont --consensus --racon --racon_iter 3` on test/sample_h1.fastq through the CLI (draft, the sequence after every
polishing iteration) in the shipped mode and in the reference-order mode (one graph per cluster / window, no trimming; tools/r04_consensus_deviation.py),
so that (a) the HIP library is compared with the oracle on the polished sequence of real reads, byte for byte, and (b) a change of a build rule that moves
the real-read consensus shows up as a diff of this file.       python oracle/make_golden_consensus.py
"""
import json, os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_lib import load_oracle, GOLD
from ngspeciesid_amd import cli, fastpath


def run(extra):
    out = tempfile.mkdtemp()
    args = cli.build_parser().parse_args(["--ont", "--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", out, "--t", "1", "--consensus", "--racon", "--racon_iter", "3"] + extra)
    args.k, args.w = 13, 20
    fastpath.main(args, api=load_oracle())
    refs = sorted(f for f in os.listdir(out) if f.startswith("consensus_reference_"))
    assert len(refs) == 1
    cid = refs[0][len("consensus_reference_"):-len(".fasta")]
    rec = {"c_id": int(cid), "draft": open(os.path.join(out, refs[0])).read().split("\n")[1]}
    for i in range(3):
        rec["it%d" % i] = open(os.path.join(out, "racon_cl_id_%s" % cid, "racon_polished_it_%d.fasta" % i)).read().split("\n")[:2]
    rec["consensus_fasta"] = open(os.path.join(out, "racon_cl_id_%s" % cid, "consensus.fasta")).read()
    shutil.rmtree(out)
    return rec


if __name__ == "__main__":
    rec = {"_what": __doc__.split("\n\n")[1], "shipped": run([])}
    json.dump(rec, open(os.path.join(GOLD, "sample_h1_consensus_oracle.json"), "w"), indent=1)
    print({k: (len(v[1]) if isinstance(v, list) else v if k == "c_id" else len(v)) for k, v in rec["shipped"].items()})
