"""GPU: a consensus / polishing call dealt to several contexts of the device (_capi.Api lanes: two host threads, two HIP streams) returns, group by group, what one
context returns - and what the oracle returns."""
import numpy as np
import pytest
from ngspeciesid_amd import _capi
from ngspeciesid_amd._capi import ReadSet, poa_params, polish_params, POA_LOCAL, NgsidError
from test_consensus_oracle import make_set

pytestmark = pytest.mark.gpu


def _groups(rd, rs, sizes):
    spc = rd["species"].numpy(); order = np.argsort(spc, kind="stable").astype(np.uint32)
    first = np.concatenate(([0], np.cumsum(np.bincount(spc))))
    lists = [order[first[g]:first[g] + sizes[g]] for g in range(len(sizes))]
    off = np.concatenate(([0], np.cumsum([len(x) for x in lists]))).astype(np.uint64)
    return np.concatenate(lists).astype(np.uint32), off, ReadSet.from_strings([rs.get(int(x[0]))[0] for x in lists])          # raw reads as backbones: every iteration changes them


def test_lanes_return_what_one_context_returns(gpu_api, oracle, monkeypatch):
    sp, rd, rs = make_set(1000, L=640, mu=13.0, seed=31, nsp=5)
    sizes = [min(s, int((rd["species"].numpy() == g).sum())) for g, s in enumerate([150, 60, 190, 90, 120])]
    ro, off, bb = _groups(rd, rs, sizes)
    dev = gpu_api.upload_reads(rs)
    monkeypatch.setattr(_capi, "LANE_MIN_READS", 0)
    saved = gpu_api.lanes
    try:
        out = {}
        for lanes in (1, 2, 3):
            gpu_api.lanes = lanes
            prm = polish_params(iters=2, k=13, w=20, tile_depth=4, band=0, trim=2)
            out[lanes] = (gpu_api.polish(bb, dev, off, prm, read_order=ro), gpu_api.polish_trace(bb, dev, off, prm, read_order=ro, aln=True),
                          gpu_api.poa_consensus(dev, off, poa_params(mode=POA_LOCAL, tile_depth=4, band=0, trim=1), read_order=ro),
                          gpu_api.polish(bb, dev, off, polish_params(iters=3, k=13, w=20, tile_depth=4, band=0, trim=2, stop_when_stable=1, single_below=64), read_order=ro))
        assert len(gpu_api.contexts()) == 3                                     # the lanes ran in contexts of their own
        for lanes in (2, 3):
            a, b = out[1], out[lanes]
            assert a[0][0] == b[0][0] and np.array_equal(a[0][1], b[0][1])
            assert a[1][0] == b[1][0] and np.array_equal(a[1][1], b[1][1]) and np.array_equal(a[1][2], b[1][2])
            assert a[2] == b[2]
            assert a[3][0] == b[3][0] and np.array_equal(a[3][1], b[3][1])
        want, used = oracle.polish(bb, rs, off, polish_params(iters=2, k=13, w=20, tile_depth=4, band=0, trim=2), read_order=ro)
        assert out[2][0][0] == want and np.array_equal(out[2][0][1], used)
        # an error in a lane's group reaches the caller (a base outside ACGTN in a read of the LAST group: dealt to the second context)
        gpu_api.lanes = 2
        bad = rs.seq.copy(); bad[int(rs.off[int(ro[-1])]) + 5] = ord("!")
        dbad = gpu_api.upload_reads(ReadSet(bad, rs.qual, rs.off))
        with pytest.raises(NgsidError):
            gpu_api.polish(bb, dbad, off, polish_params(iters=1, k=13, w=20, tile_depth=4), read_order=ro)
        dbad.release()
    finally:
        gpu_api.lanes = saved
        dev.release()


def test_one_group_or_few_reads_stay_in_one_context(gpu_api):
    sp, rd, rs = make_set(300, L=500, mu=14.0, seed=5, nsp=2)
    dev = gpu_api.upload_reads(rs)
    try:
        assert gpu_api._lane_deal(dev, [0, 150, 300]) is None                   # below LANE_MIN_READS
        assert gpu_api._lane_deal(rs, [0, 150, 300]) is None                    # host buffers: every lane would upload them again
    finally:
        dev.release()


def test_a_context_of_ones_own_closes_its_lane_contexts(monkeypatch):
    """runtime.new_api() contexts have no lanes unless asked (their makers run pipelines of their own); with lanes their twins are made on first use and go with close()"""
    from ngspeciesid_amd import runtime
    sp, rd, rs = make_set(400, L=500, mu=14.0, seed=8, nsp=2)
    ro, off, bb = _groups(rd, rs, [150, 150])
    monkeypatch.setattr(_capi, "LANE_MIN_READS", 0)
    prm = polish_params(iters=1, k=13, w=20, tile_depth=4, band=0, trim=2)
    with runtime.new_api() as a:
        assert a.lanes == 1
        dev = a.upload_reads(rs)
        one = a.polish(bb, dev, off, prm, read_order=ro)
        assert len(a.contexts()) == 1
        a.lanes = 2
        two = a.polish(bb, dev, off, prm, read_order=ro)
        assert len(a.contexts()) == 2 and one[0] == two[0] and np.array_equal(one[1], two[1])
        a.set_option("poa_host_levels", 1)                                      # an option reaches the lane contexts as well
        assert a.polish(bb, dev, off, prm, read_order=ro)[0] == one[0]
        dev.release()
    assert a.ctx is None and a.contexts() == [a]


@pytest.mark.parametrize("t", [1, 4])
def test_cli_writes_the_same_files_with_and_without_lanes(gpu_api, tmp_path, monkeypatch, t):
    """file in -> files out (draft, three polishing iterations, the PAF of every iteration): byte-identical whether the consensus calls run in one context or in two;
    --t 4: the batches of a clustering round run side by side in the two contexts as well (fastpath.cluster fn_many), the per-round dumps included"""
    import os
    from ngspeciesid_amd import synth, fastio
    from ngspeciesid_amd.cli import cli
    from test_gpu_cli import _files
    sp = synth.make_species(4, 700, 0.15, seed=3)
    rd = synth.make_reads(sp, 12000, mu=15.0, seed=4, abundance=[0.4, 0.3, 0.2, 0.1])
    rs = ReadSet(rd["seq"].numpy(), rd["qual"].numpy(), rd["off"].numpy().astype(np.uint64))
    names = fastio.Names.from_list(["r%d" % i for i in range(rs.n)])
    fq = str(tmp_path / "in.fastq"); fastio.write_fastq(fq, np.arange(rs.n), names, rs)
    monkeypatch.setattr(_capi, "LANE_MIN_READS", 0)
    saved = gpu_api.lanes; res = {}
    try:
        for lanes in (1, 2):
            gpu_api.lanes = lanes; out = str(tmp_path / ("out%d" % lanes))
            cli(["--ont", "--fastq", fq, "--outfolder", out, "--t", str(t), "--consensus", "--racon", "--racon_iter", "3", "--abundance_ratio", "0.02", "--polish_all_iterations"])
            res[lanes] = _files(out)
    finally:
        gpu_api.lanes = saved
    assert sorted(res[1]) == sorted(res[2]) and any(f.endswith("read_alignments_it_2.paf") for f in res[1])
    for f in res[1]: assert res[1][f] == res[2][f], f
    assert len([f for f in res[1] if f.endswith("consensus.fasta")]) == 4
    assert (t == 1) == (not any(f.startswith("1/") for f in res[1]))
    assert len(gpu_api.contexts()) >= 2


def test_calls_from_short_lived_threads(gpu_api, oracle):
    """a context driven from a fresh host thread per call (a server's request threads): the library's pinned staging vectors are per thread, their blocks come from and go back to
    a process-wide cache (no hipHostFree, which waits for the device, at thread exit) - same bytes from every thread, and the same as from this one"""
    import threading
    sp, rd, rs = make_set(600, L=520, mu=13.0, seed=17, nsp=2)
    ro, off, bb = _groups(rd, rs, [250, 250])
    prm = polish_params(iters=2, k=13, w=20, tile_depth=4, band=0, trim=2)
    want = gpu_api.polish(bb, rs, off, prm, read_order=ro)
    got = []
    def work(): got.append(gpu_api.polish(bb, rs, off, prm, read_order=ro))
    for _ in range(4):
        th = threading.Thread(target=work); th.start(); th.join()
    assert len(got) == 4
    for g in got: assert g[0] == want[0] and np.array_equal(g[1], want[1])
    assert want[0] == oracle.polish(bb, rs, off, prm, read_order=ro)[0]
