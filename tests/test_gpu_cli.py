"""GPU: the CLI / host layer end to end on the reference's own test file, against the reference's output files."""
import hashlib, os
import numpy as np
import pytest
from oracle_lib import GOLD
from test_host_parallelize import run_tree

pytestmark = pytest.mark.gpu


def test_cli_sample_h1_t1(gpu_api, tmp_path):
    from ngspeciesid_amd.cli import cli
    out = str(tmp_path / "out")
    cli(["--ont", "--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", out, "--t", "1"])
    assert hashlib.md5(open(os.path.join(out, "sorted.fastq"), "rb").read()).hexdigest() == open(os.path.join(GOLD, "sample_h1_t1_sorted.fastq.md5")).read().strip()
    assert open(os.path.join(out, "final_clusters.tsv")).read() == open(os.path.join(GOLD, "sample_h1_t1_final_clusters.tsv")).read()
    got = [l.rstrip("\n").split("\t") for l in open(os.path.join(out, "final_cluster_origins.tsv"))]
    exp = [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLD, "sample_h1_t1_final_cluster_origins.tsv"))]
    assert len(got) == len(exp)
    for a, b in zip(got, exp):
        assert a[:5] == b[:5]
        assert abs(float(a[5]) - float(b[5])) <= 16 * np.spacing(float(b[5]))       # error-rate column: summation order


def test_cli_consensus_racon(gpu_api, tmp_path):
    """--consensus --racon --racon_iter 3 on sample_h1 (BASELINE config[0]): runs through, writes the reference's output files."""
    from ngspeciesid_amd.cli import cli
    out = str(tmp_path / "out")
    cli(["--ont", "--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", out, "--t", "1", "--consensus", "--racon", "--racon_iter", "3"])
    refs = [f for f in os.listdir(out) if f.startswith("consensus_reference_")]
    assert len(refs) == 1                                       # the fw and rc clusters (138 + 115 reads) are merged by detect_reverse_complements
    cid = refs[0][len("consensus_reference_"):-len(".fasta")]
    hdr, seq = open(os.path.join(out, refs[0])).read().split("\n")[:2]
    assert hdr == ">consensus_cl_id_%s_total_supporting_reads_253" % cid
    pol = open(os.path.join(out, "racon_cl_id_%s" % cid, "consensus.fasta")).read().split("\n")[1]
    assert 600 < len(pol) < 720 and set(pol) <= set("ACGTN")
    assert os.path.exists(os.path.join(out, "reads_to_consensus_%s.fastq" % cid))


@pytest.mark.parametrize("tag,t", [("sample_h1", 8), ("synth2k_d15", 8), ("synth600_d10_q14", 4)])
def test_tree_merge(gpu_api, tag, t):
    run_tree(gpu_api, tag, t)
