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
    """--consensus --racon --racon_iter 3 on sample_h1 (BASELINE config[0]) through the HIP library: the reference's output files, and the draft and the
    sequence after EVERY polishing iteration equal, byte for byte, what the oracle returns on these real reads (tests/golden/sample_h1_consensus_oracle.json,
    oracle/make_golden_consensus.py; tests/test_cli_fastpath_cpu.py pins the oracle itself on the same file)."""
    import json
    from ngspeciesid_amd.cli import cli
    out = str(tmp_path / "out")
    cli(["--ont", "--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", out, "--t", "1", "--consensus", "--racon", "--racon_iter", "3"])
    gold = json.load(open(os.path.join(GOLD, "sample_h1_consensus_oracle.json")))["shipped"]
    refs = [f for f in os.listdir(out) if f.startswith("consensus_reference_")]
    assert len(refs) == 1                                       # the fw and rc clusters (138 + 115 reads) are merged by detect_reverse_complements
    cid = refs[0][len("consensus_reference_"):-len(".fasta")]
    assert int(cid) == gold["c_id"]
    hdr, seq = open(os.path.join(out, refs[0])).read().split("\n")[:2]
    assert hdr == ">consensus_cl_id_%s_total_supporting_reads_253" % cid
    assert seq == gold["draft"]
    folder = os.path.join(out, "racon_cl_id_%s" % cid)
    for i in range(3):                                          # run_racon's per-iteration files (consensus.py:112-120)
        assert open(os.path.join(folder, "racon_polished_it_%d.fasta" % i)).read().split("\n")[:2] == gold["it%d" % i]
        assert os.path.exists(os.path.join(folder, "racon_stderr_it_%d.txt" % i)) and os.path.exists(os.path.join(folder, "mm2_stderr_it_%d.txt" % i))
    assert open(os.path.join(folder, "consensus.fasta")).read() == gold["consensus_fasta"]
    assert os.path.exists(os.path.join(out, "reads_to_consensus_%s.fastq" % cid))
    # round 6: the file set of the folder is the reference's list (consensus.py:112-125), incl. minimap2's PAF of every iteration - and the PAF bytes are the oracle backend's
    want = sorted(["stdout.txt", "consensus.fasta"] + [f.format(i) for i in range(3) for f in ("read_alignments_it_{0}.paf", "mm2_stderr_it_{0}.txt", "racon_stderr_it_{0}.txt", "racon_polished_it_{0}.fasta")])
    assert sorted(os.listdir(folder)) == want
    from ngspeciesid_amd import cli as _cli, fastpath
    from oracle_lib import load_oracle
    out2 = str(tmp_path / "out_oracle"); os.makedirs(out2)
    args = _cli.build_parser().parse_args(["--ont", "--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", out2, "--t", "1", "--consensus", "--racon", "--racon_iter", "3"]); args.k, args.w = 13, 20
    fastpath.main(args, api=load_oracle())
    for i in range(3):
        a = open(os.path.join(folder, "read_alignments_it_%d.paf" % i)).read(); b = open(os.path.join(out2, "racon_cl_id_%s" % cid, "read_alignments_it_%d.paf" % i)).read()
        assert a == b and a.count("\n") > 200


def test_consensus_deviation_from_the_reference_order_mode(gpu_api):
    """the shipped mode (tiles of pipeline.TILE_DEPTH = 4 reads since round 5 - 6 before -, trimmed tile consensuses, one-third rule) against the mode that restates the reference's tools (ONE
    graph per cluster / window in file order, spoa's untrimmed bundle, racon's window rule), POLISHED vs POLISHED, on the reference's own reads and on
    C3-shaped clusters.  The numbers are those of profiles/r04_consensus_deviation.json (same tool, 2 000 reads per cluster, oracle backend); here 400 reads
    per cluster at mu = 14 on the HIP library.  On synthetic reads the shipped mode returns the amplicon and every difference between the modes is an error
    of the reference-order mode; on sample_h1 (no truth known) the two polished sequences differ by 12 interior edits + 22 bases of end overhang (tiles of 6: 10 + 21;
    the two tile depths differ from each other by 9 interior edits - 91 reads at 13.6 % error do not determine the sequence to the base).
    Round 5 took the reference-order mode apart rule by rule on the oracle (profiles/r05_reference_order.json, tests/test_consensus_oracle.py::
    test_reference_order_rules_of_round_5): its errors are junction insertions that come from source / sink nodes at the window edges, which the two
    racon rules this build replaces do not remove; the HIP library implements rule set 0 = the mode asserted here."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import r04_consensus_deviation as D
    from ngspeciesid_amd import synth
    from ngspeciesid_amd._capi import ReadSet
    from util_seq import edit_distance, overlap_distance
    L = open(os.path.join(GOLD, "sample_h1.fastq")).read().split("\n")
    rs = ReadSet.from_strings([L[i + 1] for i in range(0, len(L) - 3, 4)], [L[i + 3] for i in range(0, len(L) - 3, 4)])
    A = D.run(gpu_api, rs, 0.1, D.SHIPPED); B = D.run(gpu_api, rs, 0.1, D.REFORDER)
    assert len(A) == len(B) == 1 and A[0][0] == 253
    assert edit_distance(A[0][2], B[0][2]) == 34 and overlap_distance(A[0][2], B[0][2]) == 12          # the recorded deviation on real reads (oracle == HIP)
    sp = synth.make_species(5, 750, 0.15, seed=1)
    rd = synth.make_reads(sp, 2000, mu=14.0, seed=11)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    truths = [s.tobytes().decode() for s in sp]
    A = D.run(gpu_api, rs, 0.02, D.SHIPPED); B = D.run(gpu_api, rs, 0.02, D.REFORDER)
    assert len(A) == len(B) == 5
    assert sorted(c[2] for c in A) == sorted(truths)                                                   # shipped: every polished sequence IS its amplicon
    tot = 0
    for c in B:
        d = D.best_ed(c[2], truths); tot += d
        assert d <= 12                                          # reference-order mode: up to 7 edits per 750 bases at 400 reads (window junctions, untrimmed ends)
    print("reference-order mode vs truth: %d edits over 5 clusters" % tot)


def test_cli_exact_order_draft_option(gpu_api, oracle, tmp_path):
    """--poa_tile_depth 0 (extension flag): the draft of a cluster is ONE graph built in read order (spoa's order) instead of the depth-6 hierarchy; runs
    through the CLI on the HIP library and gives the oracle backend's bytes, file for file (the accuracy comparison of the two shapes is
    test_consensus_deviation_from_the_reference_order_mode)."""
    from ngspeciesid_amd import cli as _cli, fastpath
    res = {}
    for name, api in (("hip", gpu_api), ("oracle", oracle)):
        out = str(tmp_path / ("out0_" + name))
        args = _cli.build_parser().parse_args(["--ont", "--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", out, "--t", "1", "--consensus", "--max_seqs_for_consensus", "120", "--poa_tile_depth", "0"])
        args.k, args.w = 13, 20
        os.makedirs(out, exist_ok=True)
        fastpath.main(args, api=api)
        res[name] = _files(out)
    assert sorted(res["hip"]) == sorted(res["oracle"])
    refs = sorted(f for f in res["hip"] if f.startswith("consensus_reference_"))
    assert len(refs) == 1
    for k in res["hip"]:
        if k != "logfile.txt": assert res["hip"][k] == res["oracle"][k], k
    seq = res["hip"][refs[0]].decode().split("\n")[1]
    assert 600 < len(seq) < 720 and set(seq) <= set("ACGT")


def _files(out):
    res = {}
    for root, _, fs in os.walk(out):
        for f in fs:
            res[os.path.relpath(os.path.join(root, f), out)] = open(os.path.join(root, f), "rb").read()
    return res


@pytest.mark.parametrize("extra", [["--t", "1", "--consensus", "--racon", "--racon_iter", "3"], ["--t", "8", "--consensus", "--racon", "--racon_iter", "2", "--abundance_ratio", "0.01"],
                                   ["--t", "2", "--m", "620", "--s", "40", "--top_reads", "--sample_size", "150"]])
def test_cli_array_path_equals_reference_shaped_layer(gpu_api, tmp_path, monkeypatch, extra):
    """the array path (default) and the dict / file layer that mirrors the reference's Python functions write byte-identical files (HIP backend)"""
    from ngspeciesid_amd.cli import cli, build_parser
    import dict_layer
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    cli(["--ont", "--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", a] + extra)
    os.makedirs(b)
    args = build_parser().parse_args(["--ont", "--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", b] + extra); args.k, args.w = 13, 20
    dict_layer.run(args)
    fa, fb = _files(a), _files(b)
    assert sorted(fa) == sorted(fb)
    for k in fa:
        assert fa[k] == fb[k], k


def test_cli_synthetic_100k_consensus_equals_amplicons(gpu_api, tmp_path):
    """file in -> files out at C2 scale with 5 species: every racon_cl_id_*/consensus.fasta equals a generating amplicon, final_clusters.tsv is pure"""
    from ngspeciesid_amd import synth, fastio
    from ngspeciesid_amd._capi import ReadSet
    from ngspeciesid_amd.cli import cli
    sp = synth.make_species(5, 750, 0.15, seed=1)
    rd = synth.make_reads(sp, 100000, mu=17.0, seed=9)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    spc = rd["species"].numpy()
    names = fastio.Names.from_list(["r%d_sp%d" % (i, spc[i]) for i in range(rs.n)])
    fq = str(tmp_path / "in.fastq"); fastio.write_fastq(fq, np.arange(rs.n), names, rs)
    out = str(tmp_path / "out")
    cli(["--ont", "--fastq", fq, "--outfolder", out, "--t", "1", "--consensus", "--racon", "--racon_iter", "3", "--abundance_ratio", "0.02"])
    truths = sorted(s.tobytes().decode() for s in sp)
    got = sorted(open(os.path.join(out, d, "consensus.fasta")).read().split("\n")[1] for d in os.listdir(out) if d.startswith("racon_cl_id_"))
    assert got == truths
    cl = {}
    for line in open(os.path.join(out, "final_clusters.tsv")):
        cid, acc = line.rstrip("\n").split("\t"); cl.setdefault(int(cid), set()).add(acc.rsplit("_sp", 1)[1])
    assert all(len(v) == 1 for k, v in cl.items() if k < 5)


@pytest.mark.parametrize("tag,t", [("sample_h1", 8), ("synth2k_d15", 8), ("synth600_d10_q14", 4), ("synth1200_k25", 2), ("synth1200_k30", 2)])
def test_tree_merge(gpu_api, tag, t):
    run_tree(gpu_api, tag, t)


@pytest.mark.parametrize("tag,t", [("sample_h1", 8), ("synth2k_d15", 8), ("synth600_d10_q14", 4)])
def test_c_merge_representatives(gpu_api, tag, t):
    from test_host_parallelize import run_round1_then_c_merge
    run_round1_then_c_merge(gpu_api, tag, t)


def test_cli_primer_and_tail_trimming_hip_equals_oracle_backend(gpu_api, oracle, tmp_path):
    """(f4) --remove_universal_tails and --primer_file (primers with IUPAC codes) through libngsid_hip.so on the GPU box: every
    output file equals the run with the oracle as the backend byte for byte, both strands present (rc merge inside the trimming flow), and the
    reported consensus is the amplicon body cut where the reference cuts (barcode_trimmer.py:84-98)."""
    import argparse
    from ngspeciesid_amd import synth, fastio, barcode_trimmer, cli as _cli, fastpath
    from ngspeciesid_amd._capi import ReadSet
    tails = barcode_trimmer.get_universal_tails()
    bodies = synth.make_species(2, 520, 0.15, seed=8)
    amps = [np.frombuffer((tails["1_F_fw"] + b.tobytes().decode() + tails["2_R_fw"]).encode(), dtype=np.uint8) for b in bodies]
    rd = synth.make_reads(amps, 1600, mu=16.0, seed=3, rc_fraction=0.5)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    fq = str(tmp_path / "in.fastq"); fastio.write_fastq(fq, np.arange(rs.n), fastio.Names.from_list(["r%d" % i for i in range(rs.n)]), rs)
    # primer file with IUPAC codes: R = A/G, Y = C/T, N = any (edlib additionalEqualities, barcode_trimmer.py:9-12)
    f, r = tails["1_F_fw"], tails["2_R_rc"]
    iu = lambda s: "".join({"A": "R", "C": "Y"}.get(c, c) if i % 5 == 2 else ("N" if i == 7 else c) for i, c in enumerate(s))
    pf = tmp_path / "primers.fa"; pf.write_text(">F\n%s\n>R\n%s\n" % (iu(f), iu(r)))
    for extra in (["--remove_universal_tails"], ["--primer_file", str(pf)]):
        res = {}
        for name, api in (("hip", gpu_api), ("oracle", oracle)):
            out = str(tmp_path / ("out_" + name + str(len(extra))))
            args = _cli.build_parser().parse_args(["--ont", "--fastq", fq, "--outfolder", out, "--t", "1", "--consensus", "--racon", "--racon_iter", "2", "--abundance_ratio", "0.05"] + extra)
            args.k, args.w = 13, 20
            os.makedirs(out, exist_ok=True)
            fastpath.main(args, api=api)
            res[name] = _files(out)
        assert sorted(res["hip"]) == sorted(res["oracle"])
        for k in res["hip"]:
            assert res["hip"][k] == res["oracle"][k], k
        cons = sorted(v.decode().split("\n")[1] for k, v in res["hip"].items() if k.startswith("racon_cl_id_") and k.endswith("consensus.fasta"))
        # fw-oriented centre: the start cut keeps the last base of 1_F_fw; rc-oriented centre (the rc merge keeps the larger cluster's strand): 2_R_rc leads
        want = set(f[-1] + b.tobytes().decode() for b in bodies) | set(r[-1] + _rc(b.tobytes().decode()) for b in bodies)
        assert len(cons) == 2 and all(c in want for c in cons), [len(c) for c in cons]


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))
