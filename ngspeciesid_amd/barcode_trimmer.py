"""(f4) Primer / universal-tail trimming of the consensus sequences - the reference's modules/barcode_trimmer.py behind the same names
(read_barcodes, get_universal_tails, find_barcode_locations, remove_barcodes; NGSpeciesID:134-152), with edlib's infix alignment replaced by
the library's bit-parallel locator (ngsid_host_infix_locate, csrc/host_io.hip).  The work is a few dozen 150-base windows per run."""
from __future__ import annotations
import ctypes as C
import logging
from . import runtime
from .help_functions import readfq
from .consensus import reverse_complement


def read_barcodes(primer_file):
    """{name_fw: primer, name_rc: reverse complement of the upper-cased primer}   (barcode_trimmer.py:14-22)"""
    with open(primer_file) as f:
        barcodes = {acc + "_fw": seq.strip() for acc, (seq, _) in readfq(f)}
    for acc, seq in list(barcodes.items()):
        barcodes[acc[:-3] + "_rc"] = reverse_complement(seq.upper())
    return barcodes


def get_universal_tails():
    barcodes = {"1_F_fw": "TTTCTGTTGGTGCTGATATTGC", "2_R_rc": "ACTTGCCTGTCGCTCTATCTTC"}
    barcodes["1_F_rc"] = reverse_complement(barcodes["1_F_fw"])
    barcodes["2_R_fw"] = reverse_complement(barcodes["2_R_rc"])
    return barcodes


def infix_locate(query, target, max_ed, iupac=True, lib=None, prefix="ngsid_"):
    """-> (edit distance, start, end inclusive) of the first best infix alignment of query in target, or None above max_ed (edlib HW, task=locations, [0])"""
    lib = lib or runtime.load_library()
    q = query.encode(); t = target.encode()
    ed, st, en = C.c_int32(), C.c_int32(), C.c_int32()
    rc = getattr(lib, prefix + "host_infix_locate")(q, C.c_int32(len(q)), t, C.c_int32(len(t)), C.c_int32(int(max_ed)), C.c_int32(int(iupac)), C.byref(ed), C.byref(st), C.byref(en))
    if rc:
        raise RuntimeError("ngsid_host_infix_locate failed (%d)" % rc)
    return None if ed.value < 0 else (ed.value, st.value, en.value)


def find_barcode_locations(center, barcodes, primer_max_ed, lib=None, prefix="ngsid_"):
    """[(primer name, start, end, edit distance)] for every primer found within primer_max_ed (IUPAC codes in primers allowed; barcode_trimmer.py:34-60)"""
    out = []
    for acc, seq in barcodes.items():
        r = infix_locate(seq, center, primer_max_ed, True, lib, prefix)
        logging.debug(f"{acc} {r}")
        if r is not None:
            out.append((acc, r[1], r[2], r[0]))
    return out


def remove_barcodes(centers, barcodes, args, lib=None, prefix="ngsid_"):
    """cuts the consensus sequences (centers[i][2]) at the primer sites found in their first / last trim_window bases; returns True if any changed
    (barcode_trimmer.py:63-104, including its cut positions: the start cut keeps the primer's last base, the end cut is at the primer's first base)"""
    updated = False
    for i, c in enumerate(centers):
        center = c[2]
        tw = len(center) // 2 if 2 * args.trim_window > len(center) else args.trim_window
        beg = find_barcode_locations(center[:tw], barcodes, args.primer_max_ed, lib, prefix)
        end = find_barcode_locations(center[-tw:], barcodes, args.primer_max_ed, lib, prefix) if tw > 0 else []
        cut_start = 0
        for bc, start, stop, ed in beg:
            if stop > cut_start:
                cut_start = stop
        cut_end = len(center)
        if end:
            earliest = len(center)
            for bc, start, stop, ed in end:
                if start < earliest:
                    earliest = start
            cut_end = len(center) - (tw - earliest)
        if cut_start > 0 or cut_end < len(center):
            centers[i][2] = center[cut_start:cut_end]
            logging.debug(f"cut start {cut_start} cut end {cut_end}")
            updated = True
    return updated
