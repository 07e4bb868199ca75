"""CPU: the host ingest / egress helpers of the library (csrc/host_io.hip through ngspeciesid_amd.fastio) against the package's general reader
(which follows the reference's readfq) and against plain Python formatting.  No GPU needed: these entry points do no device work."""
import os
import numpy as np
import pytest
from oracle_lib import GOLD
from ngspeciesid_amd import fastio
from ngspeciesid_amd.help_functions import readfq


def _general(path):
    with open(path) as f:
        return [(a, s, q) for a, (s, q) in readfq(f)]


def test_read_fastq_equals_general_reader():
    path = os.path.join(GOLD, "sample_h1.fastq")
    names, rs, plain = fastio.read_fastq(path)
    ref = _general(path)
    assert plain and rs.n == len(ref) == len(names)
    for i, (a, s, q) in enumerate(ref):
        assert names.get(i) == a and rs.get(i) == (s, q)


def test_irregular_files_use_the_general_reader(tmp_path):
    p = tmp_path / "multi.fastq"
    p.write_text("@r1 desc\nACGT\nAC\n+\nIIII\nII\n@r2\nGG\n+\n##\n")                 # multi-line record
    names, rs, plain = fastio.read_fastq(str(p))
    assert not plain and [names.get(i) for i in range(2)] == ["r1 desc", "r2"] and rs.get(0) == ("ACGTAC", "IIIIII") and rs.get(1) == ("GG", "##")
    p2 = tmp_path / "nonl.fastq"
    p2.write_text("@a\nACGT\n+\nIIII\n@b\nTT\n+\n!!")                                     # no trailing newline: still the array path
    names, rs, plain = fastio.read_fastq(str(p2))
    assert plain and rs.n == 2 and rs.get(1) == ("TT", "!!") and names.get(1) == "b"
    p3 = tmp_path / "plusq.fastq"
    p3.write_text("@a\nACGT\n+\n@III\n@b\nTT\n+\n+!\n")                                   # quality strings starting with '@' / '+'
    names, rs, plain = fastio.read_fastq(str(p3))
    assert plain and rs.get(0) == ("ACGT", "@III") and rs.get(1) == ("TT", "+!")
    p4 = tmp_path / "empty.fastq"; p4.write_text("")
    names, rs, plain = fastio.read_fastq(str(p4))
    assert rs.n == 0


def test_writers(tmp_path):
    names, rs, _ = fastio.read_fastq(os.path.join(GOLD, "sample_h1.fastq"))
    idx = np.array([5, 0, 279, 17], dtype=np.uint64)
    out = str(tmp_path / "o.fastq")
    fastio.write_fastq(out, idx, names, rs, suffixes=["_1.5", "_22.0", "", "_x"])
    exp = "".join("@%s%s\n%s\n+\n%s\n" % (names.get(int(i)), sfx, *rs.get(int(i))) for i, sfx in zip(idx, ["_1.5", "_22.0", "", "_x"]))
    assert open(out).read() == exp
    fastio.write_fastq(out, idx[:2], names, rs, first_token=True, append=True)
    exp += "".join("@%s\n%s\n+\n%s\n" % (names.get(int(i)).split()[0], *rs.get(int(i))) for i in idx[:2])
    assert open(out).read() == exp
    tsv = str(tmp_path / "o.tsv")
    fastio.write_tsv(tsv, idx, names, fastio.int_prefixes([0, 0, 7, 1234567]))
    assert open(tsv).read() == "".join("%d\t%s\n" % (p, names.get(int(i))) for p, i in zip([0, 0, 7, 1234567], idx))
    b, o = fastio.int_prefixes(np.array([0, 9, 10, 99, 100, 12345678901]))
    assert [b[int(o[i]):int(o[i + 1])].tobytes().decode() for i in range(6)] == ["0", "9", "10", "99", "100", "12345678901"]


def test_normalize_bases():
    a = np.frombuffer(b"ACGTNacgtnRYKMUu-*xACGT", dtype=np.uint8).copy()
    ch = fastio.normalize_bases(a)
    assert a.tobytes() == b"ACGTNACGTNNNNNNNNNNACGT" and ch == 14


def test_repr_doubles_is_cpythons_repr():
    """the '_score' suffix of sorted.fastq is "{0}".format(float): the native formatter must print every double exactly like CPython"""
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.random(40000) * 700, rng.random(20000) * 1e-6, rng.random(20000) * 1e20, 10.0 ** rng.integers(-30, 30, 20000) * rng.random(20000),
                           rng.integers(0, 10 ** 17, 20000).astype(np.float64), np.frombuffer(rng.bytes(8 * 60000), dtype=np.float64),
                           np.array([0.0, -0.0, 1.0, -1.5, 1e16, 1e15, 9999999999999998.0, 1e-4, 1e-5, 0.0001234, 123456789012345678.0, float("inf"), -float("inf"), float("nan"),
                                     5e-324, 1.7976931348623157e308, 100.0, 0.1, 1 / 3])])
    b, o = fastio.repr_doubles(vals, prefix="_")
    got = [b[int(o[i]):int(o[i + 1])].tobytes().decode() for i in range(len(vals))]
    assert got == ["_" + repr(v) for v in vals.tolist()]
    b, o = fastio.repr_doubles([2.5, 1e22])
    assert b.tobytes().decode() == "2.51e+22" and o.tolist() == [0, 3, 8]


def test_count_foreign_bases():
    a = np.frombuffer(b"ACGTNacgtRYACGT", dtype=np.uint8).copy()
    assert fastio.count_foreign_bases(a) == 6


def test_malformed_short_records_are_rejected_with_many_threads(tmp_path, monkeypatch):
    """with records shorter than 64 bytes the record pass used more threads than the line pass; its per-thread error array was
    overrun and a malformed record ('-' for '+') went unnoticed.  1 M 11-byte records, 32 host threads, one bad record near the end."""
    monkeypatch.setenv("NGSID_HOST_THREADS", "32")
    n = 1000000
    rec = b"@r\nAC\n+\nII\n"
    buf = bytearray(rec * n)
    bad_at = (n - 7) * len(rec) + 6
    assert buf[bad_at:bad_at + 1] == b"+"
    buf[bad_at:bad_at + 1] = b"-"
    p = tmp_path / "bad.fastq"; p.write_bytes(bytes(buf))
    import ctypes as C
    from ngspeciesid_amd import runtime
    lib = runtime.load_library()
    a = np.frombuffer(bytes(buf), dtype=np.uint8)
    cnt = C.c_uint64(0)
    assert lib.ngsid_host_fastq_index(fastio._p(a), C.c_uint64(len(a)), None, None, None, C.c_uint64(0), C.byref(cnt)) == 0 and cnt.value == n
    recs = np.zeros(4 * n, dtype=np.uint64); nl = np.zeros(n, dtype=np.uint32); sl = np.zeros(n, dtype=np.uint32)
    assert lib.ngsid_host_fastq_index(fastio._p(a), C.c_uint64(len(a)), fastio._p(recs), fastio._p(nl), fastio._p(sl), C.c_uint64(n), C.byref(cnt)) == 1
    assert int((sl == 2).sum()) == n - 1          # every other record was still indexed
    good = rec * 50000
    a2 = np.frombuffer(good, dtype=np.uint8)
    assert lib.ngsid_host_fastq_index(fastio._p(a2), C.c_uint64(len(a2)), fastio._p(recs), fastio._p(nl), fastio._p(sl), C.c_uint64(n), C.byref(cnt)) == 0 and cnt.value == 50000


def test_crlf_fastq_reads_like_text_mode(tmp_path):
    """a CRLF FASTQ kept a trailing '\\r' on every name, sequence and quality line in the array path; the reference (and this package's
    general reader) open the file in text mode, which drops it."""
    txt = "@r1 some description\r\nACGTACGT\r\n+\r\nIIIIHHHH\r\n@r2\r\nGGTT\r\n+r2\r\n##!!\r\n@r3\r\nA\r\n+\r\nI"      # last line without a terminator
    p = tmp_path / "crlf.fastq"; p.write_bytes(txt.encode())
    names, rs, plain = fastio.read_fastq(str(p))
    ref = _general(str(p))
    assert plain and rs.n == 3 == len(ref)
    for i, (a, s, q) in enumerate(ref):
        assert names.get(i) == a and rs.get(i) == (s, q), (i, names.get(i), rs.get(i))
    assert rs.get(0) == ("ACGTACGT", "IIIIHHHH") and names.get(0) == "r1 some description" and rs.get(2) == ("A", "I")
    p2 = tmp_path / "crlf_end.fastq"; p2.write_bytes((txt + "\r\n").encode())
    names, rs, plain = fastio.read_fastq(str(p2))
    assert plain and rs.get(2) == ("A", "I")


def test_argsort_desc_is_the_stable_descending_sort():
    """ngsid_host_argsort_desc == sorted(range(n), key=score, reverse=True) (get_sorted_fastq_for_cluster.py:174): ties keep the input order"""
    rng = np.random.default_rng(5)
    cases = [np.zeros(0), np.array([3.5]), rng.random(100000) * 300.0, np.round(rng.random(200000) * 40.0, 1),                 # many ties
             np.concatenate([rng.standard_normal(5000) * 1e-300, [0.0, -0.0, 0.0, np.inf, -np.inf, 1e308, -1e308, 5e-324, -5e-324]]),
             np.full(70000, 7.25), np.array([1.0, np.nan, 2.0, np.nan, 2.0, -1.0])]
    for v in cases:
        got = fastio.argsort_desc(v)
        want = np.array(sorted(range(len(v)), key=lambda i: (v[i] != v[i], -v[i] if v[i] == v[i] else 0.0)), dtype=np.int64)   # NaNs last, else descending, stable
        assert np.array_equal(got, want)
        if not np.isnan(v).any():
            assert np.array_equal(got, np.argsort(-v, kind="stable"))


def test_list_positions_counts_earlier_reads_of_the_cluster():
    rng = np.random.default_rng(6)
    n = 50000
    rep = np.minimum(np.arange(n), rng.integers(0, 40, n)).astype(np.int64)
    rep[rep > 0] = np.where(rng.random((rep > 0).sum()) < 0.1, np.arange(n)[rep > 0], rep[rep > 0])           # some singletons
    pos = fastio.list_positions(rep)
    seen = {}
    for i, r in enumerate(rep.tolist()):
        assert pos[i] == seen.get(r, 0); seen[r] = seen.get(r, 0) + 1
    assert len(fastio.list_positions(np.zeros(0, dtype=np.int64))) == 0
    with pytest.raises(ValueError):
        fastio.list_positions(np.array([0, 5], dtype=np.int64))


def test_native_background_writers_equal_the_synchronous_ones(tmp_path):
    """round 5: ngsid_host_write_records_async - the record writer as a background job of the library (native worker threads, no interpreter thread): same bytes as the
    synchronous call for FASTQ (suffix per record / per read, first token) and TSV, many jobs in flight at once, the failure of one job reported after ALL were waited for"""
    import ctypes as C
    from ngspeciesid_amd import runtime
    names, rs, plain = fastio.read_fastq(os.path.join(GOLD, "sample_h1.fastq"))
    rng = np.random.default_rng(5)
    jobs = fastio.NativeJobs(); want = {}
    for j in range(12):
        idx = rng.permutation(rs.n)[: int(rng.integers(1, rs.n))]
        sfx = ["_%d" % int(x) for x in rng.integers(0, 10 ** int(rng.integers(1, 9)), len(idx))]
        a, b = str(tmp_path / ("a%d.fq" % j)), str(tmp_path / ("b%d.fq" % j))
        fastio.write_fastq(a, idx, names, rs, suffixes=sfx, first_token=bool(j % 2))
        fastio.write_fastq(b, idx, names, rs, suffixes=sfx, first_token=bool(j % 2), jobs=jobs)
        want[b] = a
    pre = fastio.int_prefixes(rng.integers(0, 100000, rs.n))
    fastio.write_tsv(str(tmp_path / "a.tsv"), np.arange(rs.n), names, pre)
    fastio.write_tsv(str(tmp_path / "b.tsv"), np.arange(rs.n), names, pre, jobs=jobs); want[str(tmp_path / "b.tsv")] = str(tmp_path / "a.tsv")
    assert len(jobs.jobs) == 13
    jobs.wait()
    assert jobs.jobs == []
    for b, a in want.items():
        assert open(b, "rb").read() == open(a, "rb").read(), b
    fastio.write_fastq(str(tmp_path / "no_such_dir" / "x.fq"), np.arange(3), names, rs, jobs=jobs)
    fastio.write_fastq(str(tmp_path / "ok.fq"), np.arange(3), names, rs, jobs=jobs)
    with pytest.raises(OSError, match="no_such_dir"):
        jobs.wait()
    assert os.path.getsize(str(tmp_path / "ok.fq")) > 0                       # the other job of the batch was completed
    assert runtime.load_library().ngsid_host_async_wait(C.c_uint64(123456789)) != 0      # unknown job id


def test_int_prefixes_are_the_decimal_strings():
    rng = np.random.default_rng(9)
    v = np.concatenate([[0, 9, 10, 99, 100, 2 ** 62, -1, -10, -(2 ** 63)], rng.integers(0, 10 ** 7, 20000), rng.integers(-1000, 1000, 500)]).astype(np.int64)
    buf, off = fastio.int_prefixes(v)
    got = [buf[int(off[i]):int(off[i + 1])].tobytes().decode() for i in range(len(v))]
    assert got == [str(int(x)) for x in v]
    assert fastio.int_prefixes(np.zeros(0, dtype=np.int64))[1].tolist() == [0]


def test_library_maps_and_unmaps_the_input_file(tmp_path):
    """round 5: ngsid_host_map_file / ngsid_host_unmap_file (the ingest's mapping, released off the launch thread): the mapped bytes are the file's, an empty file maps to
    nothing, a missing file is an error; read_fastq leaves its mapping to release_mapped() / the next read_fastq and the arrays it returned stay valid after the release"""
    import ctypes as C
    from ngspeciesid_amd import runtime
    lib = runtime.load_library()
    src = os.path.join(GOLD, "sample_h1.fastq"); raw = open(src, "rb").read()
    ptr = C.POINTER(C.c_ubyte)(); n = C.c_uint64(0)
    assert lib.ngsid_host_map_file(os.fsencode(src), C.byref(ptr), C.byref(n)) == 0 and n.value == len(raw)
    assert bytes(np.ctypeslib.as_array(ptr, shape=(n.value,))) == raw
    assert lib.ngsid_host_unmap_file(ptr, n, C.c_int32(0)) == 0
    empty = tmp_path / "empty.fq"; empty.write_bytes(b"")
    assert lib.ngsid_host_map_file(os.fsencode(str(empty)), C.byref(ptr), C.byref(n)) == 0 and n.value == 0
    assert lib.ngsid_host_map_file(os.fsencode(str(tmp_path / "missing.fq")), C.byref(ptr), C.byref(n)) != 0
    with pytest.raises(OSError):
        fastio.read_fastq(str(tmp_path / "missing.fq"))
    names, rs, plain = fastio.read_fastq(src)
    assert plain and len(fastio._MAPPED) == 1
    seq0 = rs.seq.copy(); nm0 = names.get(0)
    fastio.release_mapped()
    assert fastio._MAPPED == []
    import time; time.sleep(0.2)                                  # the background unmap has happened: nothing returned points into the mapping
    assert np.array_equal(rs.seq, seq0) and names.get(0) == nm0
    names2, rs2, _ = fastio.read_fastq(src)                       # the next call releases what the previous one left
    names3, rs3, _ = fastio.read_fastq(src)
    assert len(fastio._MAPPED) == 1 and np.array_equal(rs3.seq, seq0)
    fastio.release_mapped()
