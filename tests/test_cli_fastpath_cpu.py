"""CPU: the array path of the CLI (ngspeciesid_amd.fastpath) against the reference-shaped dict / file layer of this package and against the
reference's own output files, with the oracle as the C-ABI backend (the same comparison runs on the HIP library in tests/test_gpu_cli.py)."""
import hashlib, os, shutil, tempfile
import numpy as np
import pytest
from oracle_lib import GOLD
from ngspeciesid_amd import cli, fastpath
import dict_layer


def _run(api, extra, shaped, fastq=None, monkeypatch=None):
    out = tempfile.mkdtemp()
    args = cli.build_parser().parse_args(["--ont", "--fastq", fastq or os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", out] + extra)
    args.k, args.w = 13, 20
    if shaped:
        dict_layer.run(args)
    else:
        fastpath.main(args, api=api)
    files = {}
    for root, _, fs in os.walk(out):
        for f in fs:
            files[os.path.relpath(os.path.join(root, f), out)] = open(os.path.join(root, f), "rb").read()
    shutil.rmtree(out)
    return files


@pytest.fixture()
def oracle_backend(oracle, monkeypatch):
    from ngspeciesid_amd import runtime
    monkeypatch.setattr(runtime, "get_api", lambda device=None: oracle)          # the dict layer asks runtime for its backend
    return oracle


@pytest.mark.parametrize("extra", [["--t", "1"], ["--t", "8"], ["--t", "1", "--consensus", "--racon", "--racon_iter", "2"],
                                   ["--t", "4", "--consensus", "--racon", "--racon_iter", "1", "--abundance_ratio", "0.01"],
                                   ["--t", "2", "--m", "620", "--s", "40", "--top_reads", "--sample_size", "150"],
                                   ["--t", "1", "--consensus", "--poa_tile_depth", "0", "--max_seqs_for_consensus", "40", "--poa_band", "128"]])
def test_array_path_writes_the_same_files_as_the_dict_layer(oracle_backend, extra):
    a = _run(oracle_backend, extra, True); b = _run(oracle_backend, extra, False)
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k] == b[k], k
    if extra == ["--t", "1"]:                                  # and the reference's own files (tests/golden, written by the reference CLI)
        assert b["final_clusters.tsv"] == open(os.path.join(GOLD, "sample_h1_t1_final_clusters.tsv"), "rb").read()
        assert hashlib.md5(b["sorted.fastq"]).hexdigest() == open(os.path.join(GOLD, "sample_h1_t1_sorted.fastq.md5")).read().strip()
    if extra[:2] == ["--t", "4"]:                              # the per-round dumps of --t 4 (parallelize.py:85-104,193) as the reference wrote them
        import json
        gold = json.load(open(os.path.join(GOLD, "sample_h1_t4_round_dumps.json")))
        assert sorted(k for k in b if k[0].isdigit()) == sorted(gold)
        for k in gold:
            if k.endswith("pre_clusters.csv"):
                assert b[k].decode() == gold[k], k
            else:                                              # (the error-rate column: summed in hash order there)
                assert [l.split("\t")[:5] for l in b[k].decode().splitlines()] == [l.split("\t")[:5] for l in gold[k].splitlines()]


def test_soft_masked_iupac_and_overlong_reads_do_not_abort(oracle_backend, tmp_path, caplog):
    """a lower-case / IUPAC base or one over-long read used to abort the run after sorted.fastq was written.  Now: such bases are
    clustered as upper case / N (the files keep the original letters) and reads beyond NGSID_MAX_READ_LEN stay singletons, both with a warning."""
    src = open(os.path.join(GOLD, "sample_h1.fastq")).read().split("\n")
    recs = [src[i:i + 4] for i in range(0, len(src) - 3, 4)]
    recs[3][1] = recs[3][1][:50].lower() + recs[3][1][50:]                       # soft-masked prefix
    recs[7][1] = recs[7][1][:20] + "RYKM" + recs[7][1][24:]                      # IUPAC codes
    long_seq = (recs[0][1] * 120)[:70000]                                         # beyond NGSID_MAX_READ_LEN = 65 535 (round 5; 16 384 before)
    recs.append(["@too_long_read", long_seq, "+", "I" * len(long_seq)])
    fq = tmp_path / "in.fastq"; fq.write_text("\n".join("\n".join(r) for r in recs) + "\n")
    import logging
    with caplog.at_level(logging.WARNING):
        files = _run(oracle_backend, ["--t", "1"], False, fastq=str(fq))
    assert "stay singletons" in caplog.text and "clustered as upper case / N" in caplog.text
    srt = files["sorted.fastq"].decode()
    assert recs[3][1] in srt and recs[7][1] in srt and long_seq in srt            # original letters in the output
    lines = files["final_clusters.tsv"].decode().splitlines()
    ids = {l.split("\t")[1]: int(l.split("\t")[0]) for l in lines}
    assert sum(1 for l in lines if int(l.split("\t")[0]) == ids["too_long_read"]) == 1          # a singleton
    ref = _run(oracle_backend, ["--t", "1"], False)
    big = lambda f: sorted(np.bincount([int(l.split(b"\t")[0]) for l in f["final_clusters.tsv"].splitlines()]).tolist(), reverse=True)[:2]
    assert big(files) == big(ref)                                                  # the masked / IUPAC reads still join their clusters


def test_unsupported_k_is_refused_before_any_output(tmp_path):
    out = tmp_path / "o"
    with pytest.raises(SystemExit):
        cli.cli(["--fastq", os.path.join(GOLD, "sample_h1.fastq"), "--outfolder", str(out), "--k", "40", "--w", "45"])
    assert not os.path.exists(out / "sorted.fastq")


def test_universal_tail_trimming_both_layers(oracle_backend, tmp_path):
    """(f4) --remove_universal_tails / --primer_file: amplicons flanked by the universal tails; the polished consensus is the amplicon body, cut where the reference cuts (the start cut keeps the last primer base, barcode_trimmer.py:84-98)"""
    from ngspeciesid_amd import synth, fastio, barcode_trimmer
    from ngspeciesid_amd._capi import ReadSet
    tails = barcode_trimmer.get_universal_tails()
    body = synth.make_species(1, 420, 0.15, seed=8)[0]
    amp = np.frombuffer((tails["1_F_fw"] + body.tobytes().decode() + tails["2_R_fw"]).encode(), dtype=np.uint8)
    rd = synth.make_reads([amp], 260, mu=18.0, seed=3)
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    fq = str(tmp_path / "in.fastq"); fastio.write_fastq(fq, np.arange(rs.n), fastio.Names.from_list(["r%d" % i for i in range(rs.n)]), rs)
    pf = tmp_path / "primers.fa"; pf.write_text(">F\n%s\n>R\n%s\n" % (tails["1_F_fw"], tails["2_R_rc"]))
    for extra in (["--remove_universal_tails"], ["--primer_file", str(pf)]):
        flags = ["--t", "1", "--consensus", "--racon", "--racon_iter", "2", "--abundance_ratio", "0.1"] + extra
        b = _run(oracle_backend, flags, False, fastq=fq)
        ref = [v for k, v in b.items() if k.startswith("consensus_reference_")]
        assert len(ref) == 1 and ref[0].decode().split("\n")[1] == tails["1_F_fw"][-1] + body.tobytes().decode()      # the (trimmed) draft is exact already
        cons = [v for k, v in b.items() if k.startswith("racon_cl_id_") and k.endswith("consensus.fasta")]
        assert len(cons) == 1
        hdr, seq = cons[0].decode().split("\n")[:2]
        assert seq == tails["1_F_fw"][-1] + body.tobytes().decode()
        # the trimmed sequence is what every file of the last iteration reports, with consistent header tags
        assert " LN:i:%d RC:i:" % len(seq) in hdr and hdr.endswith("XC:f:1.000000")
        last = [v for k, v in b.items() if k.startswith("racon_cl_id_") and k.endswith("racon_polished_it_1.fasta")]
        assert last == cons


def test_background_writers_and_synchronous_writes_give_the_same_files(oracle_backend, monkeypatch):
    """the output files are complete when main() returns, with the writer threads (default) and without (NGSID_CLI_SYNC_WRITES=1)"""
    flags = ["--t", "2", "--consensus", "--racon", "--racon_iter", "1", "--abundance_ratio", "0.01"]
    a = _run(oracle_backend, flags, False)
    monkeypatch.setenv("NGSID_CLI_SYNC_WRITES", "1")
    b = _run(oracle_backend, flags, False)
    assert sorted(a) == sorted(b) and all(a[k] == b[k] for k in a)
    assert any(k.startswith("reads_to_consensus_") for k in a) and "sorted.fastq" in a


def test_polished_sample_h1_is_pinned(oracle_backend):
    """regression pin of the consensus half on the reference's own reads: draft, the sequence after every polishing iteration and consensus.fasta of
    `--consensus --racon --racon_iter 3` (oracle backend) equal tests/golden/sample_h1_consensus_oracle.json - NOT a reference vector (spoa / racon are
    absent), the file a rule change has to move; tests/test_gpu_cli.py compares the HIP library with the same file."""
    import json
    gold = json.load(open(os.path.join(GOLD, "sample_h1_consensus_oracle.json")))["shipped"]
    f = _run(oracle_backend, ["--t", "1", "--consensus", "--racon", "--racon_iter", "3"], False)
    cid = gold["c_id"]
    assert f["consensus_reference_%d.fasta" % cid].decode().split("\n")[1] == gold["draft"]
    for i in range(3):
        assert f["racon_cl_id_%d/racon_polished_it_%d.fasta" % (cid, i)].decode().split("\n")[:2] == gold["it%d" % i]
        assert "racon_cl_id_%d/racon_stderr_it_%d.txt" % (cid, i) in f and "racon_cl_id_%d/mm2_stderr_it_%d.txt" % (cid, i) in f
    assert f["racon_cl_id_%d/consensus.fasta" % cid].decode() == gold["consensus_fasta"]



@pytest.mark.parametrize("sync_writes", [False, True])
def test_centres_that_merge_only_after_trimming_are_polished_again(oracle_backend, tmp_path, monkeypatch, sync_writes):
    """The second pass of the trimming flow (NGSpeciesID:147-152: when a primer is still found after polishing, the reference runs detect_reverse_complements +
    polish_sequences again).  Round 5: the flow has the reference's order - trim the drafts, polish the trimmed sequences with clipped reads, look again - so on clean data
    the second look finds nothing and there is no second pass; both are forced here (the second remove_barcodes reports a removal, the second merge decision joins the two
    centres).  Checked: the folders and files of the absorbed centre are gone, the surviving centre's files carry the pooled reads of both clusters, the supporting-read
    count in every header is the sum, and the reported sequence is the trimmed amplicon body."""
    from ngspeciesid_amd import synth, fastio, barcode_trimmer, pipeline
    from ngspeciesid_amd._capi import ReadSet
    tails = barcode_trimmer.get_universal_tails()
    bodies = synth.make_species(2, 400, 0.2, seed=12)
    amps = [np.frombuffer((tails["1_F_fw"] + b.tobytes().decode() + tails["2_R_fw"]).encode(), dtype=np.uint8) for b in bodies]
    rd = synth.make_reads(amps, 330, mu=19.0, seed=4, abundance=np.array([0.8, 0.2]))
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    fq = str(tmp_path / "in.fastq"); fastio.write_fastq(fq, np.arange(rs.n), fastio.Names.from_list(["r%d" % i for i in range(rs.n)]), rs)
    real = pipeline.detect_reverse_complements
    calls = []

    def forced(api, centers, thr):
        calls.append(len(centers))
        out = real(api, centers, thr)
        if len(calls) >= 2 and len(out) == 2:                 # the second decision (second pass): the smaller centre joins the larger one
            a, b = out
            return [[a[0] + b[0], a[1], a[2], list(a[3]) + list(b[3])]]
        return out

    monkeypatch.setattr(pipeline, "detect_reverse_complements", forced)
    real_rm = barcode_trimmer.remove_barcodes
    rm_calls = []

    def rm(centers, barcodes, args):
        rm_calls.append(len(centers))
        found = real_rm(centers, barcodes, args)
        return True if len(rm_calls) == 2 else found          # the look after polishing "finds" a primer: the reference's second pass runs

    monkeypatch.setattr(barcode_trimmer, "remove_barcodes", rm)
    if sync_writes: monkeypatch.setenv("NGSID_CLI_SYNC_WRITES", "1")          # (both writer modes - background threads that start before the draft, and inline writes)
    else: monkeypatch.delenv("NGSID_CLI_SYNC_WRITES", raising=False)
    flags = ["--t", "1", "--consensus", "--racon", "--racon_iter", "2", "--abundance_ratio", "0.05", "--remove_universal_tails"]
    files = _run(oracle_backend, flags, False, fastq=fq)
    assert calls == [2, 2] and rm_calls == [2, 2], (calls, rm_calls)          # merge on the trimmed drafts, look after polishing, merge of the second pass
    refs = sorted(k for k in files if k.startswith("consensus_reference_"))
    folders = sorted({k.split("/")[0] for k in files if k.startswith("racon_cl_id_")})
    pooled = sorted(k for k in files if k.startswith("reads_to_consensus_"))
    assert len(refs) == 1 and len(folders) == 1, (refs, folders)
    cid = refs[0][len("consensus_reference_"):-len(".fasta")]
    assert folders == ["racon_cl_id_" + cid] and ("reads_to_consensus_%s.fastq" % cid) in pooled
    # (the absorbed centre's reads_to_consensus_* file of the FIRST pass stays on disk, as in the reference: polish_sequences removes the racon folders and the
    #  consensus_reference_* files before a pass, consensus.py:195-200, not the pooled read files)
    assert len(pooled) == 2
    clusters = [l.split("\t")[0] for l in files["final_clusters.tsv"].decode().splitlines()]
    n0, n1 = clusters.count("0"), clusters.count("1")
    assert n0 > n1 > 15
    assert files["reads_to_consensus_%s.fastq" % cid].count(b"\n+\n") == n0 + n1
    pooled_names = [l[1:].split()[0] for l in files["reads_to_consensus_%s.fastq" % cid].decode().split("\n")[0::4] if l]
    member_names = [l.split("\t")[1] for l in files["final_clusters.tsv"].decode().splitlines() if l.split("\t")[0] in ("0", "1")]
    assert sorted(n.rsplit("_", 1)[0] for n in pooled_names) == sorted(member_names)          # (the pooled files carry the accession WITH its `_score` suffix, the TSV without: NGSpeciesID:104-106) the rewritten content: every read of both clusters, once
    hdr, seq = files["racon_cl_id_%s/consensus.fasta" % cid].decode().split("\n")[:2]
    assert hdr.startswith(">consensus_cl_id_%s_total_supporting_reads_%d " % (cid, n0 + n1)) and " LN:i:%d " % len(seq) in hdr
    assert seq == tails["1_F_fw"][-1] + bodies[0].tobytes().decode()          # polished by the pooled reads (4 : 1 its own), trimmed like the reference trims
    assert files["racon_cl_id_%s/racon_polished_it_1.fasta" % cid] == files["racon_cl_id_%s/consensus.fasta" % cid]


def test_writer_modes_leave_the_same_files(oracle_backend, monkeypatch):
    """round 5: the record writers as background jobs of the library (default), as interpreter threads (NGSID_CLI_PY_WRITERS=1, rounds 2 - 5) and synchronous
    (NGSID_CLI_SYNC_WRITES=1): the same file set with the same bytes, consensus stage and rc merge included"""
    extra = ["--t", "1", "--consensus", "--racon", "--racon_iter", "2", "--abundance_ratio", "0.02", "--rc_identity_threshold", "0.9"]
    native = _run(oracle_backend, extra, False)
    monkeypatch.setenv("NGSID_CLI_PY_WRITERS", "1")
    threads = _run(oracle_backend, extra, False)
    monkeypatch.delenv("NGSID_CLI_PY_WRITERS"); monkeypatch.setenv("NGSID_CLI_SYNC_WRITES", "1")
    sync = _run(oracle_backend, extra, False)
    assert sorted(native) == sorted(threads) == sorted(sync)
    assert any(k.startswith("reads_to_consensus_") for k in native) and "final_clusters.tsv" in native and "sorted.fastq" in native
    for k in native:
        assert native[k] == threads[k] == sync[k], k


def test_a_concatemer_in_an_abundant_cluster_is_left_out_of_the_consensus(oracle_backend, tmp_path, caplog):
    """a read longer than MAX_CONSENSUS_LEN (13 107 bases) that joins an abundant cluster used to end the run with a ValueError after the cluster files were
    written.  Now it stays in final_clusters.tsv and in the pooled read file, the draft and the polisher leave it out (warning), and the consensus is the one of the run without it."""
    import logging
    src = open(os.path.join(GOLD, "sample_h1.fastq")).read().split("\n")
    recs = [src[i:i + 4] for i in range(0, len(src) - 3, 4)]
    base = _run(oracle_backend, ["--t", "1", "--consensus", "--racon", "--racon_iter", "1"], False)
    unit = recs[0][1]; n = fastpath.MAX_CONSENSUS_LEN // len(unit) + 1
    recs.append(["@concatemer", unit * n, "+", recs[0][3] * n])
    assert len(recs[-1][1]) > fastpath.MAX_CONSENSUS_LEN
    fq = tmp_path / "in.fastq"; fq.write_text("\n".join("\n".join(r) for r in recs) + "\n")
    with caplog.at_level(logging.WARNING):
        files = _run(oracle_backend, ["--t", "1", "--consensus", "--racon", "--racon_iter", "1"], False, fastq=str(fq))
    assert "left out of the consensus" in caplog.text
    cl = [l.split("\t") for l in files["final_clusters.tsv"].decode().splitlines()]
    cid = [c for c, a in cl if a == "concatemer"][0]
    assert sum(1 for c, a in cl if c == cid) > 50                                   # it sits in an abundant cluster
    pooled = [k for k in files if k.startswith("reads_to_consensus_") and b"@concatemer_" in files[k]]
    assert len(pooled) == 1
    cons = sorted(v.split(b"\n")[1] for k, v in files.items() if k.endswith("consensus.fasta"))
    assert len(cons) == len([k for k in base if k.endswith("consensus.fasta")]) and all(600 < len(c) < 720 for c in cons)


def test_racon_folder_holds_the_reference_file_list_incl_the_paf(oracle_backend):
    """run_racon (consensus.py:107-126) leaves, per iteration i, read_alignments_it_{i}.paf, mm2_stderr_it_{i}.txt, racon_stderr_it_{i}.txt and
    racon_polished_it_{i}.fasta, + stdout.txt and consensus.fasta.  The PAF comes from the records of ngsid_polish_trace_aln: 12 columns, one line per aligned read of the pooled
    file, coordinates inside the sequences, the target = the sequence the iteration started from; --skip_paf leaves it out."""
    files = _run(oracle_backend, ["--t", "1", "--consensus", "--racon", "--racon_iter", "2"], False)
    folder = sorted({k.split("/")[0] for k in files if k.startswith("racon_cl_id_")})
    assert len(folder) == 1
    got = sorted(k.split("/", 1)[1] for k in files if k.startswith(folder[0] + "/"))
    want = sorted(["stdout.txt", "consensus.fasta"] + [f.format(i) for i in range(2) for f in ("read_alignments_it_{0}.paf", "mm2_stderr_it_{0}.txt", "racon_stderr_it_{0}.txt", "racon_polished_it_{0}.fasta")])
    assert got == want
    cid = folder[0][len("racon_cl_id_"):]
    pooled = files["reads_to_consensus_%s.fastq" % cid].decode().split("\n")
    qlen = {pooled[i][1:]: len(pooled[i + 1]) for i in range(0, len(pooled) - 3, 4)}
    start = files["consensus_reference_%s.fasta" % cid].decode().split("\n")[1]
    it0 = files[folder[0] + "/racon_polished_it_0.fasta"].decode().split("\n")[1]
    for i, target in enumerate((start, it0)):
        lines = [l.split("\t") for l in files[folder[0] + "/read_alignments_it_%d.paf" % i].decode().splitlines()]
        assert 200 < len(lines) <= len(qlen) and all(len(l) == 12 for l in lines)
        assert len({l[0] for l in lines}) == len(lines)                                   # one line per read
        for l in lines:
            ql, qs, qe, tl, ts, te, nm, bl = (int(l[x]) for x in (1, 2, 3, 6, 7, 8, 9, 10))
            assert ql == qlen[l[0]] and 0 <= qs < qe <= ql and tl == len(target) and 0 <= ts < te <= tl
            assert l[4] in "+-" and l[5] == "consensus_cl_id_%s_total_supporting_reads_253" % cid and l[11] == "255"
            assert bl == max(qe - qs, te - ts) and 0 <= nm <= bl and nm > 0.6 * bl        # reads of the cluster at ~13 % error
        assert {l[4] for l in lines} == {"+", "-"}                                         # the merged centre pools the forward and the reverse-complement cluster
    skipped = _run(oracle_backend, ["--t", "1", "--consensus", "--racon", "--racon_iter", "2", "--skip_paf"], False)
    assert sorted(k for k in files if not k.endswith(".paf")) == sorted(skipped) and all(files[k] == skipped[k] for k in skipped if k != "logfile.txt")
