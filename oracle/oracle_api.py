"""ctypes access to the oracle (oracle/liboracle.so) and to the reference build (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg - never by necat_amd/ (the product)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
ORACLE_BIN = os.path.join(HERE, "oc2pmov_oracle")
REF_PMOV = os.path.join(HERE, "_ref", "oc2pmov")
REF_MKDB = os.path.join(HERE, "_ref", "oc2mkdb")
REF_LIB = os.path.join(HERE, "_ref", "libnecat_ref.so")


class OraOptions(C.Structure):
    _fields_ = [("kmer_size", C.c_int), ("scan_window", C.c_int), ("kmer_cnt_cutoff", C.c_int),
                ("block_size", C.c_int), ("block_score_cutoff", C.c_int), ("num_candidates", C.c_int),
                ("align_size_cutoff", C.c_int), ("ddfs_cutoff", C.c_double), ("error", C.c_double),
                ("num_output", C.c_int), ("num_threads", C.c_int), ("job", C.c_int),
                ("binary_output", C.c_int), ("use_hdr_as_id", C.c_int)]


class OraStats(C.Structure):
    _fields_ = [("n_records", C.c_uint64), ("aligned_qbases", C.c_uint64), ("t_index", C.c_double), ("t_map", C.c_double)]


class OraVolume(C.Structure):
    _fields_ = [("pac", C.c_void_p), ("nbases", C.c_uint64), ("nseq", C.c_uint64), ("offset", C.c_void_p),
                ("size", C.c_void_p), ("hdr_offset", C.c_void_p), ("hdr", C.c_void_p), ("hdr_bytes", C.c_uint64)]


class OraIndex(C.Structure):
    _fields_ = [("kmer_stats", C.POINTER(C.c_uint64)), ("offset_list", C.POINTER(C.c_uint64)),
                ("n_offsets", C.c_uint64), ("k", C.c_int)]


class OraAlignResult(C.Structure):
    _fields_ = [("qoff", C.c_int), ("qend", C.c_int), ("toff", C.c_int), ("tend", C.c_int),
                ("ident_perc", C.c_double), ("align_size", C.c_int),
                ("query_align", C.c_void_p), ("target_align", C.c_void_p)]


class _KString(C.Structure):
    _fields_ = [("l", C.c_size_t), ("m", C.c_size_t), ("s", C.c_char_p)]


class _RefOcAlignData(C.Structure):
    # leading fields of OcAlignData (gapped_align/oc_aligner.h:6-15): what a caller reads back
    _fields_ = [("edlib", C.c_void_p), ("qoff", C.c_int), ("qend", C.c_int), ("toff", C.c_int), ("tend", C.c_int),
                ("ident_perc", C.c_double), ("query_align", _KString), ("target_align", _KString)]


_lib = None
_ref_lib = None


def ref_lib() -> C.CDLL:
    """The reference's own libontcns subset (oracle/_ref/libnecat_ref.so, built from /root/reference by oracle/Makefile)."""
    global _ref_lib
    if _ref_lib is None:
        l = C.CDLL(REF_LIB)
        l.new_OcAlignData.argtypes = [C.c_double]
        l.new_OcAlignData.restype = C.POINTER(_RefOcAlignData)
        l.free_OcAlignData.argtypes = [C.POINTER(_RefOcAlignData)]
        l.free_OcAlignData.restype = C.c_void_p
        l.onc_align.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(_RefOcAlignData),
                                C.c_int, C.c_int, C.c_int]
        l.onc_align.restype = C.c_int8        # BOOL is int8_t (common/ontcns_defs.h:19)
        _ref_lib = l
    return _ref_lib


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
        l = C.CDLL(LIB)
        l.ora_options_default.argtypes = [C.POINTER(OraOptions)]
        l.ora_pm_main.argtypes = [C.POINTER(OraOptions), C.c_int, C.c_char_p, C.c_char_p, C.POINTER(OraStats)]
        l.ora_volume_load.argtypes = [C.c_char_p, C.POINTER(OraVolume)]
        l.ora_volume_free.argtypes = [C.POINTER(OraVolume)]
        l.ora_index_build.argtypes = [C.POINTER(OraVolume), C.c_int, C.c_int]
        l.ora_index_build.restype = C.POINTER(OraIndex)
        l.ora_index_free.argtypes = [C.POINTER(OraIndex)]
        l.ora_aligner_new.argtypes = [C.c_double]
        l.ora_aligner_new.restype = C.c_void_p
        l.ora_aligner_free.argtypes = [C.c_void_p]
        l.ora_edlib_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.ora_onc_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.POINTER(OraAlignResult)]
        _lib = l
    return _lib


def options(**kw) -> OraOptions:
    o = OraOptions()
    lib().ora_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def pm_main(opt: OraOptions, vid: int, wrk_dir: str, output: str) -> OraStats:
    st = OraStats()
    rc = lib().ora_pm_main(C.byref(opt), vid, wrk_dir.encode(), output.encode(), C.byref(st))
    if rc != 0:
        raise RuntimeError("oracle pm_main failed (%d)" % rc)
    return st


def build_index(volume_path: str, k: int, max_occ: int):
    """Returns (kmer_stats copy, offset_list copy)."""
    v = OraVolume()
    if lib().ora_volume_load(volume_path.encode(), C.byref(v)) != 0:
        raise RuntimeError("oracle cannot load " + volume_path)
    ix = lib().ora_index_build(C.byref(v), k, max_occ)
    T = 1 << (2 * k)
    stats = np.ctypeslib.as_array(ix.contents.kmer_stats, shape=(T,)).copy()
    n = int(ix.contents.n_offsets)
    offs = np.ctypeslib.as_array(ix.contents.offset_list, shape=(max(n, 1),))[:n].copy()
    lib().ora_index_free(ix)
    lib().ora_volume_free(C.byref(v))
    return stats, offs


_OPS = {"M": 0, "I": 1, "D": 2, "X": 3}


def edlib_align(query: np.ndarray, target: np.ndarray, error: float = 0.5):
    """Edlib_align on byte-coded sequences: (ok, dist, qend, tend, ops[uint8]) with the op codes of
    include/necat_hip.h (0 match, 1 ins, 2 del, 3 mismatch)."""
    l = lib()
    a = l.ora_aligner_new(error)
    q = np.ascontiguousarray(query, dtype=np.uint8)
    t = np.ascontiguousarray(target, dtype=np.uint8)
    qa = C.create_string_buffer(8192)
    ta = C.create_string_buffer(8192)
    qe, te, d = C.c_int(), C.c_int(), C.c_int()
    ok = l.ora_edlib_align(a, q.ctypes.data, q.shape[0], t.ctypes.data, t.shape[0], qa, ta, C.byref(qe), C.byref(te), C.byref(d))
    l.ora_aligner_free(a)
    ops = []
    if ok:
        for x, y in zip(qa.value, ta.value):
            if x == 45:
                ops.append(2)
            elif y == 45:
                ops.append(1)
            else:
                ops.append(0 if x == y else 3)
    return bool(ok), d.value, qe.value, te.value, np.asarray(ops, dtype=np.uint8)


class Aligner:
    """onc_align (gapped_align/oc_aligner.c:303) of the oracle, or - impl='ref' - of the reference build itself.
    align() returns (ok, qoff, qend, toff, tend, ident_perc, query_align, target_align) with the gapped
    strings as bytes ("ACGT-")."""

    def __init__(self, error: float = 0.5, impl: str = "oracle"):
        self.impl = impl
        if impl == "ref":
            self.h = ref_lib().new_OcAlignData(error)
        else:
            self.h = lib().ora_aligner_new(error)

    def align(self, query: np.ndarray, qstart: int, target: np.ndarray, tstart: int, min_align: int, tail_match_len: int,
              block_size: int = 512):
        q = np.ascontiguousarray(query, dtype=np.uint8)
        t = np.ascontiguousarray(target, dtype=np.uint8)
        if self.impl == "ref":
            ok = ref_lib().onc_align(q.ctypes.data, qstart, q.shape[0], t.ctypes.data, tstart, t.shape[0], self.h,
                                     block_size, min_align, tail_match_len)
            d = self.h.contents
            qa = C.string_at(d.query_align.s, d.query_align.l) if d.query_align.l else b""
            ta = C.string_at(d.target_align.s, d.target_align.l) if d.target_align.l else b""
            return bool(ok), d.qoff, d.qend, d.toff, d.tend, d.ident_perc, qa, ta
        r = OraAlignResult()
        ok = lib().ora_onc_align(self.h, q.ctypes.data, qstart, q.shape[0], t.ctypes.data, tstart, t.shape[0],
                                 block_size, min_align, tail_match_len, C.byref(r))
        n = r.align_size
        qa = C.string_at(r.query_align, n) if n else b""
        ta = C.string_at(r.target_align, n) if n else b""
        return bool(ok), r.qoff, r.qend, r.toff, r.tend, r.ident_perc, qa, ta

    def close(self):
        if self.h:
            if self.impl == "ref":
                ref_lib().free_OcAlignData(self.h)
            else:
                lib().ora_aligner_free(self.h)
            self.h = None


def have_ref() -> bool:
    return os.path.exists(REF_PMOV)


def opt_argv(o) -> List[str]:
    """MapOptions2String (common/map_options.c:70-87)"""
    return ["-k", str(o.kmer_size), "-z", str(o.scan_window), "-q", str(o.kmer_cnt_cutoff), "-b", str(o.block_size),
            "-s", str(o.block_score_cutoff), "-n", str(o.num_candidates), "-a", str(o.align_size_cutoff),
            "-d", "%f" % o.ddfs_cutoff, "-e", "%f" % o.error, "-m", str(o.num_output), "-t", str(o.num_threads),
            "-j", str(o.job), "-u", str(o.binary_output), "-i", str(o.use_hdr_as_id)]


def run_ref(o, vid: int, wrk_dir: str, output: str, binary: Optional[str] = None) -> float:
    """Run the compiled reference oc2pmov; returns the 'pairwise mapping' seconds summed from its log."""
    out = subprocess.run([binary or REF_PMOV] + opt_argv(o) + [wrk_dir, str(vid), output], check=True,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    t = 0.0
    for line in out.splitlines():
        if "'pairwise mapping" in line and "takes" in line:
            t += float(line.split("takes")[1].split("secs")[0])
    return t


def record_size(opt) -> int:
    """bytes per record of an oc2pmov output file, 0 = text lines: 28-byte PackedGappedCandidate (-j 0 -u 1),
    96-byte M4Record (-j 1 -u 1)"""
    if not opt.binary_output:
        return 0
    return 96 if opt.job == 1 else 28


def sorted_records(path: str, binary_size: int = 0) -> List[bytes]:
    """Records of an output file, sorted (the order threads flush in is unspecified).  The 4 padding bytes that end
    a binary M4Record (m4_record.h:10-25: int vscore, then alignment padding) are zeroed: the reference dumps an
    uninitialised stack struct there (pm_worker.c:41,79)."""
    b = open(path, "rb").read()
    if binary_size == 96:
        return sorted(b[i:i + 92] + b"\0\0\0\0" for i in range(0, len(b), 96))
    if binary_size:
        return sorted(b[i:i + binary_size] for i in range(0, len(b), binary_size))
    return sorted(b.splitlines(keepends=True))


# ---- consensus stage, extension loop (cns_oracle.c; reference side = oracle/cns_ref_harness.c) ----

class OraCnsOptions(C.Structure):
    """the CnsOptions fields the loop reads (consensus/cns_options.c:10-22 for the defaults)"""
    _fields_ = [("min_align_size", C.c_int), ("min_cov", C.c_int), ("max_cov", C.c_int), ("error", C.c_double),
                ("mapping_ratio", C.c_double), ("use_fixed_ident_cutoff", C.c_int)]


def cns_options(**kw) -> OraCnsOptions:
    o = OraCnsOptions(400, 4, 12, 0.5, 0.8, 0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


REF_PCAN = os.path.join(HERE, "_ref", "oc2pcan")
REF_CNS = os.path.join(HERE, "_ref", "cns_ref_harness")
REF_RM = os.path.join(HERE, "_ref", "oc2rm_worker")      # the reference's read-to-reference mapper (reference_mapping/rm_one_vol_main.c)
REF_OC2CNS = os.path.join(HERE, "_ref", "oc2cns")          # the reference's oc2cns itself (consensus/main.c)


def have_ref_cns() -> bool:
    return os.path.exists(REF_PCAN) and os.path.exists(REF_CNS)


def run_ref_pcan(wrk_dir: str, can_path: str, batch_size: int = 100000) -> None:
    """the reference's oc2pcan: can_path (28-byte records) -> can_path.p<i> + can_path.partitions"""
    subprocess.run([REF_PCAN, "-p", str(batch_size), "-t", "1", wrk_dir, can_path], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def cns_argv(o: OraCnsOptions) -> List[str]:
    return ["-a", str(o.min_align_size), "-x", str(o.min_cov), "-y", str(o.max_cov), "-e", "%f" % o.error,
            "-p", "%f" % o.mapping_ratio, "-u", str(o.use_fixed_ident_cutoff)]


def run_ref_cns(o: OraCnsOptions, wrk_dir: str, can_path: str, log_path: str, full: bool = False) -> None:
    """the reference's consensus driver with its add_one_align / consensus_broken calls logged"""
    subprocess.run([REF_CNS] + cns_argv(o) + [wrk_dir, can_path, log_path] + (["full"] if full else []), check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def cns_run(o: OraCnsOptions, wrk_dir: str, can_path: str, log_path: str, full: bool = False) -> None:
    """the oracle's restatement over the same files, same log format"""
    L = lib()
    L.ora_cns_run.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(OraCnsOptions), C.c_char_p, C.c_int]
    rc = L.ora_cns_run(wrk_dir.encode(), can_path.encode(), C.byref(o), log_path.encode(), int(full))
    if rc:
        raise RuntimeError("ora_cns_run failed: %d" % rc)


def parse_cns_log(path: str):
    """[(template_id, template_size, ident_cutoff, num_can, num_ovlps, ranges, overlaps)] with overlaps =
    [(toff, tend, weight, aln_size, hash_q, hash_t[, qaln, taln])]"""
    out, cur = [], []
    for ln in open(path):
        f = ln.rstrip("\n").split("\t")
        if f[0] == "A":
            cur.append((int(f[1]), int(f[2]), float(f[3]), int(f[4]), f[5], f[6]) + tuple(f[7:9]))
        else:
            nr = int(f[6])
            rg = [(int(f[7 + 2 * i]), int(f[8 + 2 * i])) for i in range(nr)]
            out.append((int(f[1]), int(f[2]), float(f[3]), int(f[4]), int(f[5]), rg, cur))
            cur = []
    return out


def fnv64(b: bytes) -> str:
    """the string hash of the consensus logs, as 16 hex digits"""
    L = lib()
    L.ora_fnv64.restype = C.c_uint64
    L.ora_fnv64.argtypes = [C.c_char_p, C.c_size_t]
    return "%016x" % L.ora_fnv64(b, len(b))

