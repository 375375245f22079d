"""Seeded synthetic ONT-like read sets and NECAT 2-bit volume files.

This is the data side of the benchmark / parity harness: there is no network
and no real dataset, so every workload is generated here from a seed.

Volume format (what `oc2mkdb` writes and `oc2pmov` reads), restated from the reference:
  * file layout           : common/packed_db.c:291-315 (pdb_dump) / :317-345 (pdb_load_pac)
      31-byte magic "ontcns_pac_header_hofuwhogfuewo" (no NUL) | u64 nseq | u64 nbases |
      nseq x {u64 offset, u64 size, u64 hdr_offset, i32 platform, 4 pad} | u64 hdr_bytes |
      NUL-separated read names | ceil(nbases/4) bytes of 2-bit bases
  * base packing          : common/ontcns_aux.h:118-119 (_set_pac/_get_pac) - base l lives in byte
      l>>2 at shift ((~l)&3)<<1, i.e. the FIRST base of a byte is in its TOP two bits.
  * directory files       : common/makedb_aux.c:36-45 (reads_info.txt = "V\\tN\\n"),
      makedb/main.c:31,104 (volume_names.txt = "path\\tread_start_id\\tread_count\\n" per volume)
  * volume cut            : makedb/main.c:8,29 - a volume is closed once its bases >= 2e9
      (`vol_size` below; tests shrink it to force several volumes).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

PAC_MAGIC = b"ontcns_pac_header_hofuwhogfuewo"
DEFAULT_VOL_SIZE = 2_000_000_000


@dataclass
class ReadSet:
    """Reads as one concatenated uint8 code array (0..3 = ACGT) plus per-read offsets."""

    codes: np.ndarray      # uint8 [nbases]
    offsets: np.ndarray    # int64 [nreads]
    sizes: np.ndarray      # int64 [nreads]
    names: List[str]

    @property
    def nreads(self) -> int:
        return int(self.sizes.shape[0])

    @property
    def nbases(self) -> int:
        return int(self.codes.shape[0])

    def read(self, i: int) -> np.ndarray:
        o = int(self.offsets[i])
        return self.codes[o:o + int(self.sizes[i])]


def _mutate(seq: np.ndarray, err: float, rng: np.random.Generator) -> np.ndarray:
    """Apply per-base errors at total rate `err`, split 1:1:1 substitution / insertion / deletion."""
    n = seq.shape[0]
    if err <= 0.0 or n == 0:
        return seq.copy()
    u = rng.random(n)
    op_sub = u < err / 3.0
    op_ins = (u >= err / 3.0) & (u < 2.0 * err / 3.0)
    op_del = (u >= 2.0 * err / 3.0) & (u < err)
    out = seq.copy()
    # substitution: add 1..3 mod 4 so the base always changes
    nsub = int(op_sub.sum())
    if nsub:
        out[op_sub] = (out[op_sub] + rng.integers(1, 4, nsub, dtype=np.uint8)) & 3
    reps = np.ones(n, dtype=np.int64)
    reps[op_del] = 0
    reps[op_ins] = 2
    idx = np.repeat(np.arange(n, dtype=np.int64), reps)
    res = out[idx]
    # the first copy of every inserted pair becomes a random base (insertion BEFORE the original)
    if op_ins.any():
        starts = np.cumsum(reps) - reps
        ins_pos = starts[op_ins]
        res[ins_pos] = rng.integers(0, 4, ins_pos.shape[0], dtype=np.uint8)
    return res


def make_genome(length: int, seed: int, repeat_frac: float = 0.0) -> np.ndarray:
    """i.i.d. uniform ACGT genome; `repeat_frac` optionally pastes copies of a 5 kb element
    (stress variant: repeats make k-mers exceed the occurrence cutoff and add false seeds)."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, length, dtype=np.uint8)
    if repeat_frac > 0.0 and length > 20000:
        unit = g[:5000].copy()
        ncopies = int(length * repeat_frac / 5000)
        for p in rng.integers(5000, length - 5000, ncopies):
            g[p:p + 5000] = unit
    return g


def make_genome_families(length: int, seed: int, families: Sequence[Sequence[int]]) -> np.ndarray:
    """i.i.d. uniform ACGT genome with repeat FAMILIES pasted in: `families` = [(unit_len, copies), ...], every family a random
    unit of its own, its copies at random places (later copies may overwrite earlier ones).  SURVEY 8d's stress variant at full
    size: with R-fold coverage a family of c copies puts ~ R c (1 - err)^k exact occurrences of each of its k-mers into a
    volume - above the -q cutoff (lookup_table.c:15-58) for many short copies, below it but far above a block's 40 seeds
    (word_finder.c:91-92) and a read's -n candidates (pm_worker.c:139-140, :168-171) for fewer, longer ones."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, length, dtype=np.uint8)
    for unit_len, copies in families:
        unit = rng.integers(0, 4, int(unit_len), dtype=np.uint8)
        for p in rng.integers(0, length - int(unit_len), int(copies)):
            g[p:p + int(unit_len)] = unit
    return g


def simulate_reads(genome_len: int = 4_600_000, coverage: float = 40.0, seed: int = 7,
                   mean_len: float = 8000.0, sd_len: float = 2400.0, min_len: int = 3000,
                   err: float = 0.12, repeat_frac: float = 0.0,
                   genome: Optional[np.ndarray] = None, families: Optional[Sequence[Sequence[int]]] = None,
                   lognormal: Optional[Sequence[float]] = None, max_len: Optional[int] = None) -> ReadSet:
    """SURVEY.md §8d generator: reads sampled uniformly, length ~ N(mean, sd) clipped to
    [min_len, G], strand 50/50, 12 % errors (1:1:1), names r<idx>_<start>_<len>_<strand>.
    `families`: the genome of make_genome_families.  `lognormal` = (median, sigma): a long-tailed length model instead of the
    normal one (ONT-like: a few reads of 100 kb and more), cut at `max_len`."""
    if genome is None:
        genome = make_genome_families(genome_len, seed, families) if families else make_genome(genome_len, seed, repeat_frac)
    G = int(genome.shape[0])
    rng = np.random.default_rng(seed + 1)
    target = coverage * G
    chunks: List[np.ndarray] = []
    sizes: List[int] = []
    names: List[str] = []
    total = 0
    i = 0
    lo = min(min_len, G)
    while total < target:
        L = int(rng.lognormal(np.log(lognormal[0]), lognormal[1])) if lognormal else int(rng.normal(mean_len, sd_len))
        if max_len:
            L = min(L, int(max_len))
        L = max(lo, min(L, G))
        start = int(rng.integers(0, G - L + 1))
        frag = genome[start:start + L]
        strand = int(rng.integers(0, 2))
        if strand:
            frag = (3 - frag)[::-1]
        r = _mutate(np.ascontiguousarray(frag), err, rng)
        chunks.append(r)
        sizes.append(int(r.shape[0]))
        names.append("r%d_%d_%d_%d" % (i, start, L, strand))
        total += L
        i += 1
    sz = np.asarray(sizes, dtype=np.int64)
    off = np.zeros_like(sz)
    if sz.shape[0] > 1:
        np.cumsum(sz[:-1], out=off[1:])
    codes = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    return ReadSet(codes=codes, offsets=off, sizes=sz, names=names)


def add_long_indels(rs: ReadSet, frac: float = 0.4, seed: int = 1, lo: int = 250, hi: int = 900) -> ReadSet:
    """A copy of `rs` in which a fraction of the reads carries one or two long indels (a stretch removed, or random bases
    put in): the overlaps the block-wise extension cannot bridge, which oc2cns -r 1 hands to its rescue pair."""
    rng = np.random.default_rng(seed)
    chunks: List[np.ndarray] = []
    for i in range(rs.nreads):
        r = rs.read(i)
        if rng.random() < frac:
            for _ in range(int(rng.integers(1, 3))):
                n = int(rng.integers(lo, hi))
                if r.shape[0] < 3 * n + 600:
                    break
                at = int(rng.integers(300, r.shape[0] - n - 300))
                if rng.random() < 0.5:
                    r = np.concatenate([r[:at], r[at + n:]])
                else:
                    r = np.concatenate([r[:at], rng.integers(0, 4, n, dtype=np.uint8), r[at:]])
        chunks.append(np.ascontiguousarray(r, dtype=np.uint8))
    sz = np.asarray([c.shape[0] for c in chunks], dtype=np.int64)
    off = np.zeros_like(sz)
    if sz.shape[0] > 1:
        np.cumsum(sz[:-1], out=off[1:])
    return ReadSet(codes=np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8), offsets=off, sizes=sz, names=list(rs.names))


def pack_2bit(codes: np.ndarray) -> np.ndarray:
    """2-bit pack, first base of each byte in the top two bits (ontcns_aux.h:118-119)."""
    n = codes.shape[0]
    pad = (-n) % 4
    c = np.concatenate([codes, np.zeros(pad, dtype=np.uint8)]) if pad else codes
    q = c.reshape(-1, 4)
    return ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).astype(np.uint8)


def unpack_2bit(pac: np.ndarray, nbases: int) -> np.ndarray:
    out = np.empty((pac.shape[0], 4), dtype=np.uint8)
    out[:, 0] = (pac >> 6) & 3
    out[:, 1] = (pac >> 4) & 3
    out[:, 2] = (pac >> 2) & 3
    out[:, 3] = pac & 3
    return out.reshape(-1)[:nbases]


def write_volume(path: str, codes: np.ndarray, sizes: Sequence[int], names: Sequence[str]) -> None:
    """Write one `vol%d` file, byte-compatible with packed_db.c:291-315 (pad bytes zero)."""
    sizes = np.asarray(sizes, dtype=np.uint64)
    nseq = int(sizes.shape[0])
    offs = np.zeros(nseq, dtype=np.uint64)
    if nseq > 1:
        np.cumsum(sizes[:-1], out=offs[1:])
    hdr = bytearray()
    hdr_offs = np.zeros(nseq, dtype=np.uint64)
    for i, nm in enumerate(names):
        hdr_offs[i] = len(hdr)
        hdr += nm.encode("ascii") + b"\0"
    info = np.zeros(nseq, dtype=np.dtype([("offset", "<u8"), ("size", "<u8"),
                                           ("hdr_offset", "<u8"), ("platform", "<i4"),
                                           ("pad", "<i4")]))
    info["offset"] = offs
    info["size"] = sizes
    info["hdr_offset"] = hdr_offs
    with open(path, "wb") as f:
        f.write(PAC_MAGIC)
        f.write(np.uint64(nseq).tobytes())
        f.write(np.uint64(codes.shape[0]).tobytes())
        f.write(info.tobytes())
        f.write(np.uint64(len(hdr)).tobytes())
        f.write(bytes(hdr))
        f.write(pack_2bit(codes).tobytes())


def read_volume(path: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray, List[str]]:
    """Return (pac bytes, offsets, sizes, names) of a volume file."""
    with open(path, "rb") as f:
        buf = f.read()
    m = len(PAC_MAGIC)
    if buf[:m] != PAC_MAGIC:
        raise ValueError("not a NECAT pac volume: %s" % path)
    nseq = int(np.frombuffer(buf, dtype="<u8", count=1, offset=m)[0])
    nb = int(np.frombuffer(buf, dtype="<u8", count=1, offset=m + 8)[0])
    p = m + 16
    info = np.frombuffer(buf, dtype=np.dtype([("offset", "<u8"), ("size", "<u8"),
                                               ("hdr_offset", "<u8"), ("platform", "<i4"),
                                               ("pad", "<i4")]), count=nseq, offset=p)
    p += 32 * nseq
    hb = int(np.frombuffer(buf, dtype="<u8", count=1, offset=p)[0])
    p += 8
    hdr = buf[p:p + hb]
    p += hb
    pac = np.frombuffer(buf, dtype=np.uint8, count=(nb + 3) // 4, offset=p)
    names = []
    for ho in info["hdr_offset"]:
        e = hdr.find(b"\0", int(ho))
        names.append(hdr[int(ho):e].decode("ascii"))
    return pac, info["offset"].astype(np.int64), info["size"].astype(np.int64), names


def write_volume_dir(wrk_dir: str, rs: ReadSet, vol_size: int = DEFAULT_VOL_SIZE) -> int:
    """oc2mkdb equivalent for an in-memory ReadSet: vol0.., volume_names.txt, reads_info.txt.
    Volume paths are written exactly as oc2mkdb does: `wrk_dir` + '/' + 'vol%d'."""
    os.makedirs(wrk_dir, exist_ok=True)
    base = wrk_dir if wrk_dir.endswith("/") else wrk_dir + "/"
    vid = 0
    start = 0
    cur = 0
    lines = []
    for i in range(rs.nreads + 1):
        close = False
        if i < rs.nreads:
            cur += int(rs.sizes[i])
            close = cur >= vol_size
            end = i + 1
        else:
            close = cur > 0 and start < rs.nreads
            end = rs.nreads
        if close and end > start:
            o0 = int(rs.offsets[start])
            o1 = int(rs.offsets[end - 1] + rs.sizes[end - 1])
            vname = base + "vol%d" % vid
            write_volume(vname, rs.codes[o0:o1], rs.sizes[start:end], rs.names[start:end])
            lines.append("%s\t%d\t%d\n" % (vname, start, end - start))
            vid += 1
            start = end
            cur = 0
    with open(base + "volume_names.txt", "w") as f:
        f.writelines(lines)
    with open(base + "reads_info.txt", "w") as f:
        f.write("%d\t%d\n" % (vid, rs.nreads))
    return vid


def cut_ranges(rs: ReadSet, cuts: Sequence[int]) -> List[Tuple[int, int]]:
    """Read ranges [start, end) of the volumes of `rs` when volume v is closed once it holds >= cuts[v] bases and the last volume
    takes the rest (oc2mkdb's rule, makedb/main.c:29, with a threshold per volume)."""
    out: List[Tuple[int, int]] = []
    start = cur = 0
    for i in range(rs.nreads):
        cur += int(rs.sizes[i])
        if len(out) < len(cuts) and cur >= int(cuts[len(out)]):
            out.append((start, i + 1))
            start = i + 1
            cur = 0
    if start < rs.nreads:
        out.append((start, rs.nreads))
    return out


def remainder_cuts(rs: ReadSet, num_volumes: int, last_frac: float = 0.4) -> List[int]:
    """thresholds for `num_volumes` volumes: equal ones and a last one of ~last_frac of their size - a project's last volume
    is a remainder, which is what makes its (reference, query) volume pairs unequal"""
    if num_volumes <= 1:
        return []
    return [int(rs.nbases / (num_volumes - 1 + last_frac))] * (num_volumes - 1)


def write_volume_dir_cuts(wrk_dir: str, rs: ReadSet, cuts: Sequence[int]) -> int:
    """Like write_volume_dir, but volume v is closed once it holds >= cuts[v] bases (the last volume takes the rest):
    volume sets of UNEQUAL sizes, the shape a real project has - its last volume is a remainder (makedb/main.c:29) - and what the
    (reference volume, query volume) pair scheduler has to balance."""
    os.makedirs(wrk_dir, exist_ok=True)
    base = wrk_dir if wrk_dir.endswith("/") else wrk_dir + "/"
    lines = []
    ranges = cut_ranges(rs, cuts)
    for vid, (a, b) in enumerate(ranges):
        o0 = int(rs.offsets[a])
        o1 = int(rs.offsets[b - 1] + rs.sizes[b - 1])
        vname = base + "vol%d" % vid
        write_volume(vname, rs.codes[o0:o1], rs.sizes[a:b], rs.names[a:b])
        lines.append("%s\t%d\t%d\n" % (vname, a, b - a))
    with open(base + "volume_names.txt", "w") as f:
        f.writelines(lines)
    with open(base + "reads_info.txt", "w") as f:
        f.write("%d\t%d\n" % (len(ranges), rs.nreads))
    return len(ranges)


def write_fasta(path: str, rs: ReadSet) -> None:
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    with open(path, "wb") as f:
        for i in range(rs.nreads):
            f.write(b">" + rs.names[i].encode("ascii") + b"\n")
            f.write(lut[rs.read(i)].tobytes())
            f.write(b"\n")


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="write a synthetic NECAT volume directory")
    ap.add_argument("wrk_dir")
    ap.add_argument("--genome", type=int, default=4_600_000)
    ap.add_argument("--coverage", type=float, default=40.0)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--err", type=float, default=0.12)
    ap.add_argument("--vol-size", type=int, default=DEFAULT_VOL_SIZE)
    ap.add_argument("--fasta", default=None)
    a = ap.parse_args()
    rs = simulate_reads(a.genome, a.coverage, a.seed, err=a.err)
    nv = write_volume_dir(a.wrk_dir, rs, a.vol_size)
    if a.fasta:
        write_fasta(a.fasta, rs)
    print("reads=%d bases=%d volumes=%d" % (rs.nreads, rs.nbases, nv))
