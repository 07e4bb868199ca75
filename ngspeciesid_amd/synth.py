"""Synthetic amplicon read sets (the build's own generator).

Species = one random ACGT root amplicon with i.i.d. substitutions (``divergence``) and ~1 % indels.
Reads: per-read mean phred ~ N(mu, 2.5) clipped >= 5, per-base phred ~ N(mean, 6) clipped to [1, 50];
each base is an error with probability 10^(-q/10): 40 % substitution, 30 % deletion, 30 % insertion.
Output is the CSR layout of include/ngsid.h (uint8 bases / phred+33 characters + uint64 offsets).
Works on CPU or on a CUDA/HIP torch device (the bench generates its 1 M reads directly in HBM).
"""
from __future__ import annotations
import numpy as np
import torch

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_species(n_species: int, length: int, divergence: float = 0.15, indel: float = 0.01, seed: int = 1):
    """Return a list of uint8 arrays (ASCII) - species 0 is the root itself mutated like the others."""
    rng = np.random.default_rng(seed)
    root = rng.integers(0, 4, size=length)
    out = []
    for s in range(n_species):
        r = np.random.default_rng(seed * 1000003 + s + 1)
        sub = r.random(length) < divergence
        sp = np.where(sub, (root + r.integers(1, 4, size=length)) % 4, root)
        keep = r.random(length) >= indel / 2.0
        ins = r.random(length) < indel / 2.0
        pieces = []
        for i in range(length):
            if keep[i]:
                pieces.append(sp[i])
            if ins[i]:
                pieces.append(r.integers(0, 4))
        out.append(_ACGT[np.asarray(pieces, dtype=np.int64)])
    return out


def reverse_complement_ascii(a: np.ndarray) -> np.ndarray:
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGTN", b"TGCAN"):
        comp[x] = y
    return comp[a[::-1]]


class _HashRng:
    """counter-based random numbers from plain integer tensor arithmetic (splitmix64 of seed / stream / element index): no torch.Generator, no hipRAND state - the
    same numbers on every device.  bench.py uses it for multi-process runs (eight processes sharing one GPU were seen to stall inside torch's own
    generator kernels before any library call); the one-GPU workloads keep the torch generator, so their read sets are those of the earlier rounds."""
    _M = (1 << 64) - 1

    def __init__(self, seed, dev):
        self.seed, self.dev, self.stream = int(seed), dev, 0

    @staticmethod
    def _s(x):      # python int -> the same 64 bits as a signed value
        x &= _HashRng._M
        return x - (1 << 64) if x >= (1 << 63) else x

    def _bits(self, n):
        self.stream += 1
        base = self._s((self.seed * 0x9E3779B97F4A7C15 + self.stream * 0xD6E8FEB86659FD93) & self._M)
        z = torch.arange(n, dtype=torch.int64, device=self.dev) * self._s(0x9E3779B97F4A7C15) + base
        z = (z ^ ((z >> 30) & ((1 << 34) - 1))) * self._s(0xBF58476D1CE4E5B9)
        z = (z ^ ((z >> 27) & ((1 << 37) - 1))) * self._s(0x94D049BB133111EB)
        return z ^ ((z >> 31) & ((1 << 33) - 1))

    def rand(self, *shape):
        n = int(np.prod(shape))
        return (((self._bits(n) >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))).to(torch.float32).reshape(shape)

    def randn(self, *shape):
        n = int(np.prod(shape))
        u1 = ((self._bits(n) >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53)); u2 = ((self._bits(n) >> 11) & ((1 << 53) - 1)).to(torch.float64) * (1.0 / (1 << 53))
        return (torch.sqrt(-2.0 * torch.log(1.0 - u1)) * torch.cos(6.283185307179586 * u2)).to(torch.float32).reshape(shape)

    def randint(self, lo, hi, shape, dtype=torch.uint8):
        n = int(np.prod(shape))
        return (lo + ((self._bits(n) >> 33) & 0x7FFFFFFF) % (hi - lo)).to(dtype).reshape(shape)

    def categorical(self, probs, n):
        cdf = torch.cumsum(probs.to(torch.float64), 0); cdf = cdf / cdf[-1]
        return torch.clamp(torch.searchsorted(cdf, self.rand(n).to(torch.float64), right=True), max=len(probs) - 1)


@torch.no_grad()
def make_reads(species, n_reads: int, mu: float = 17.0, seed: int = 7, device="cpu", abundance=None,
               rc_fraction: float = 0.0, chunk: int = 65536, rng: str = "torch"):
    """Generate reads.  Returns dict(seq, qual, off (torch uint8/uint8/int64 on `device`), species (int64),
    strand (uint8: 1 = reverse complement)).  rng = "torch" (torch.Generator: the read sets of every test and of the one-GPU bench) or "hash" (_HashRng)."""
    dev = torch.device(device)
    hr = _HashRng(seed, dev) if rng == "hash" else None
    g = None
    if hr is None:
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
    S = len(species)
    Lmax = max(len(s) for s in species)
    code = np.full((2 * S, Lmax), 0, dtype=np.uint8)
    lens = np.zeros(2 * S, dtype=np.int64)
    lut = np.zeros(256, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    for i, s in enumerate(species):
        code[i, :len(s)] = lut[s]
        lens[i] = len(s)
        rc = reverse_complement_ascii(s)
        code[S + i, :len(s)] = lut[rc]
        lens[S + i] = len(s)
    code_t = torch.from_numpy(code).to(dev)
    lens_t = torch.from_numpy(lens).to(dev)
    if abundance is None:
        probs = torch.full((S,), 1.0 / S, device=dev)
    else:
        probs = torch.tensor(abundance, dtype=torch.float32, device=dev)
        probs = probs / probs.sum()
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    seqs, quals, offs, sps, strands = [], [], [], [], []
    base_off = 0
    for c0 in range(0, n_reads, chunk):
        n = min(chunk, n_reads - c0)
        sp = hr.categorical(probs, n) if hr else torch.multinomial(probs, n, replacement=True, generator=g)
        strand = ((hr.rand(n) if hr else torch.rand(n, device=dev, generator=g)) < rc_fraction)
        tid = sp + strand.long() * S
        L = lens_t[tid]
        qbar = torch.clamp((hr.randn(n) if hr else torch.randn(n, device=dev, generator=g)) * 2.5 + mu, min=5.0)
        q = torch.clamp(torch.round((hr.randn(n, Lmax) if hr else torch.randn(n, Lmax, device=dev, generator=g)) * 6.0 + qbar[:, None]), 1, 50)
        p = torch.pow(10.0, -q / 10.0)
        err = (hr.rand(n, Lmax) if hr else torch.rand(n, Lmax, device=dev, generator=g)) < p
        typ = hr.rand(n, Lmax) if hr else torch.rand(n, Lmax, device=dev, generator=g)
        valid = torch.arange(Lmax, device=dev)[None, :] < L[:, None]
        is_sub = err & (typ < 0.4)
        is_del = err & (typ >= 0.4) & (typ < 0.7)
        is_ins = err & (typ >= 0.7)
        base = code_t[tid]
        subb = (base + (hr.randint(1, 4, (n, Lmax)) if hr else torch.randint(1, 4, (n, Lmax), device=dev, generator=g, dtype=torch.uint8))) % 4
        insb = hr.randint(0, 4, (n, Lmax)) if hr else torch.randint(0, 4, (n, Lmax), device=dev, generator=g, dtype=torch.uint8)
        emit = torch.where(is_sub, subb, base)
        cnt = (valid & ~is_del).long() + (valid & is_ins).long()
        incl = torch.cumsum(cnt, dim=1)
        rl = incl[:, -1]
        roff = torch.cumsum(rl, dim=0) - rl
        total = int(rl.sum().item())
        pos = roff[:, None] + incl - cnt
        out_s = torch.empty(total, dtype=torch.uint8, device=dev)
        out_q = torch.empty(total, dtype=torch.uint8, device=dev)
        qc = (q + 33).to(torch.uint8)
        m1 = valid & ~is_del
        out_s[pos[m1]] = acgt[emit[m1].long()]
        out_q[pos[m1]] = qc[m1]
        m2 = valid & is_ins
        p2 = pos[m2] + (~is_del[m2]).long()
        out_s[p2] = acgt[insb[m2].long()]
        out_q[p2] = qc[m2]
        seqs.append(out_s); quals.append(out_q)
        offs.append(roff + base_off)
        base_off += total
        sps.append(sp); strands.append(strand.to(torch.uint8))
    off = torch.cat(offs + [torch.tensor([base_off], device=dev, dtype=torch.int64)])
    return dict(seq=torch.cat(seqs), qual=torch.cat(quals), off=off, species=torch.cat(sps), strand=torch.cat(strands))


def reads_to_fastq(rd, path, prefix="r"):
    seq = rd["seq"].cpu().numpy(); qual = rd["qual"].cpu().numpy(); off = rd["off"].cpu().numpy(); sp = rd["species"].cpu().numpy()
    with open(path, "w") as f:
        for i in range(len(off) - 1):
            a, b = off[i], off[i + 1]
            f.write("@%s%d_sp%d\n%s\n+\n%s\n" % (prefix, i, sp[i], seq[a:b].tobytes().decode(), qual[a:b].tobytes().decode()))
