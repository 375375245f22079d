"""ctypes binding of libnecat_hip.so (include/necat_hip.h) + a Python mirror of pm_main.

The product is the C ABI and the oc2pmov program on top of it; this module only lets tests and
bench.py drive the same entry points.  No computation happens in Python, and there is no CPU
fallback: if the library or a GPU is missing every call raises.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import List, Optional, Tuple

import numpy as np

from . import build as _build
from .synth import read_volume


class MapOptions(C.Structure):
    """common/map_options.h:10-25"""
    _fields_ = [("kmer_size", C.c_int), ("scan_window", C.c_int), ("kmer_cnt_cutoff", C.c_int),
                ("block_size", C.c_int), ("block_score_cutoff", C.c_int), ("num_candidates", C.c_int),
                ("align_size_cutoff", C.c_int), ("ddfs_cutoff", C.c_double), ("error", C.c_double),
                ("num_output", C.c_int), ("num_threads", C.c_int), ("job", C.c_int),
                ("binary_output", C.c_int), ("use_hdr_as_id", C.c_int)]


class Timings(C.Structure):
    _fields_ = [("index_ms", C.c_double), ("seed_ms", C.c_double), ("extend_ms", C.c_double),
                ("myers_ms", C.c_double), ("traceback_ms", C.c_double), ("myers_launches", C.c_uint64),
                ("myers_blocks", C.c_uint64), ("myers_word_updates", C.c_uint64),
                ("myers_cells_bases", C.c_uint64), ("rounds", C.c_uint64),
                ("myersA_ms", C.c_double), ("myersA_launches", C.c_uint64), ("myersA_blocks", C.c_uint64),
                ("tracebackA_ms", C.c_double), ("myersA_big_ms", C.c_double), ("myersA_big_blocks", C.c_uint64),
                ("myers_band_words", C.c_uint64),
                ("fused_ms", C.c_double), ("fused_launches", C.c_uint64), ("fused_blocks", C.c_uint64),
                ("rc_ms", C.c_double), ("rc_launches", C.c_uint64), ("rc_blocks", C.c_uint64), ("rc_words", C.c_uint64),
                ("rc_ck_ms", C.c_double),
                ("seed_bases", C.c_uint64), ("seed_lookups", C.c_uint64), ("seed_hits", C.c_uint64), ("seed_cands", C.c_uint64)]


class ShardTimings(C.Structure):
    """necat_shard_timings"""
    _fields_ = [("index_local_ms", C.c_double), ("index_exchange_ms", C.c_double), ("index_exchange_bytes", C.c_uint64),
                ("gather_ms", C.c_double), ("gather_bytes", C.c_uint64), ("reads_local", C.c_uint64),
                ("index_sharded", C.c_uint64), ("index_plan_replicate_ms", C.c_double), ("index_plan_shard_ms", C.c_double)]


class IndexPlan(C.Structure):
    """necat_index_plan_t"""
    _fields_ = [("shard", C.c_int32), ("_pad", C.c_int32), ("replicate_ms", C.c_double), ("shard_ms", C.c_double), ("exchange_ms", C.c_double),
                ("exchange_bytes", C.c_uint64)]


HOST_ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

CANDIDATE_DTYPE = np.dtype([("qid", "<i4"), ("sid", "<i4"), ("qdir", "<i4"), ("sdir", "<i4"), ("score", "<i4"),
                            ("_pad", "<i4"), ("qbeg", "<u8"), ("qend", "<u8"), ("qsize", "<u8"),
                            ("sbeg", "<u8"), ("send", "<u8"), ("ssize", "<u8"), ("qoff", "<u8"), ("soff", "<u8")])
M4_DTYPE = np.dtype([("qid", "<i4"), ("qdir", "<i4"), ("qoff", "<u8"), ("qend", "<u8"), ("qext", "<u8"),
                     ("qsize", "<u8"), ("sid", "<i4"), ("sdir", "<i4"), ("soff", "<u8"), ("send", "<u8"),
                     ("sext", "<u8"), ("ssize", "<u8"), ("ident_perc", "<f8"), ("vscore", "<i4"), ("_pad", "<i4")])
ALIGNMENT_DTYPE = np.dtype([("ok", "<i4"), ("qoff", "<i4"), ("qend", "<i4"), ("toff", "<i4"), ("tend", "<i4"),
                            ("align_size", "<i4"), ("ident_perc", "<f8")])
ASM_ANCHOR_DTYPE = np.dtype([("qid", "<i4"), ("sid", "<i4"), ("sdir", "<i4"), ("qoff", "<i4"), ("soff", "<i4")])      # necat_asm_anchor
assert CANDIDATE_DTYPE.itemsize == 88 and M4_DTYPE.itemsize == 96 and ALIGNMENT_DTYPE.itemsize == 32
CNS_OVERLAP_DTYPE = np.dtype([("cand", "<u8"), ("qoff", "<i4"), ("qend", "<i4"), ("toff", "<i4"), ("tend", "<i4"),
                              ("align_size", "<i4"), ("ops_block", "<u4"), ("ops_off", "<u8"), ("ident_perc", "<f8"),
                              ("weight", "<f8")])
CNS_TEMPLATE_DTYPE = np.dtype([("examined", "<i4"), ("num_can", "<i4"), ("num_ovlps", "<i4"), ("_pad", "<i4"),
                               ("ident_cutoff", "<f8"), ("ovlp_begin", "<u8"), ("ovlp_end", "<u8"),
                               ("range_begin", "<u8"), ("range_end", "<u8")])
assert CNS_OVERLAP_DTYPE.itemsize == 56 and CNS_TEMPLATE_DTYPE.itemsize == 56


class CnsOptions(C.Structure):
    """necat_cns_options (include/necat_hip.h) = the CnsOptions fields the extension loop reads"""
    _fields_ = [("min_align_size", C.c_int), ("min_cov", C.c_int), ("max_cov", C.c_int), ("error", C.c_double),
                ("mapping_ratio", C.c_double), ("use_fixed_ident_cutoff", C.c_int), ("rescue_long_indels", C.c_int)]


class _CnsResult(C.Structure):
    _fields_ = [("n_templates", C.c_uint64), ("templates", C.c_void_p), ("n_overlaps", C.c_uint64), ("overlaps", C.c_void_p),
                ("n_ranges", C.c_uint64), ("ranges", C.c_void_p), ("n_ops_blocks", C.c_uint32), ("ops", C.POINTER(C.c_void_p)),
                ("n_aligned", C.c_uint64), ("n_used", C.c_uint64), ("n_rounds", C.c_uint32), ("device_ms", C.c_double),
                ("host_ms", C.c_double), ("n_rescue_tried", C.c_uint64), ("n_rescued", C.c_uint64), ("rescue_ms", C.c_double)]

ABI_VERSION = 6          # include/necat_hip.h: NECAT_ABI_VERSION

EXPORTED_SYMBOLS = [
    "necat_default_options", "necat_ctx_create", "necat_ctx_destroy", "necat_ctx_trim", "necat_last_error", "necat_device_name",
    "necat_volume_upload", "necat_volume_pack", "necat_volume_free", "necat_index_build", "necat_index_size", "necat_index_download",
    "necat_index_free", "necat_index_sparse_size", "necat_index_download_sparse", "necat_find_candidates", "necat_extend", "necat_map_pair", "necat_map_reference", "necat_onc_align_batch", "necat_asm_align_batch", "necat_asm_plan_batch",
    "necat_gapped_strings", "necat_cns_default_options", "necat_cns_load_partition", "necat_cns_extension_batch",
    "necat_cns_result_free",
    "necat_edlib_align_batch", "necat_get_timings", "necat_get_timings_sized", "necat_get_shard_timings_sized", "necat_abi_version", "necat_free", "necat_pcan_partition",
    "necat_comm_create", "necat_comm_destroy", "necat_comm_transport", "necat_get_shard_timings", "necat_comm_selftest_rccl", "necat_comm_selftest_rccl2",
    "necat_index_build_sharded", "necat_index_plan", "necat_find_candidates_sharded", "necat_map_pair_sharded",
    "necat_pair_schedule", "necat_pair_chunk_reads", "necat_find_candidates_part", "necat_map_pair_part",
]

_lib = None
_lib_xcheck = None


def needs_xcheck(env=None) -> bool:
    """True when the kernel-path knobs in `env` (default: os.environ) select a path of the CROSS-CHECK build - a kernel family the default paths replaced, kept as an
    independent implementation for the parity tests (necat_hip.hip, NECAT_BUILD_CROSSCHECK; libnecat_hip_xcheck.so).  The product library refuses such a knob with
    NECAT_ERR_ARG (never another path), so a wrong answer here fails a test loudly."""
    env = os.environ if env is None else env
    num = lambda k, d: int(env.get(k, d))
    rcwalk, tail = num("NECAT_RCWALK", 512), num("NECAT_TAIL_FUSED", 512)
    all_fused = tail >= 1 << 26                      # every list through k_tail_fused
    if not all_fused and (rcwalk == 0 or rcwalk > max(tail, 15)):      # list sizes no default path covers (a round's bound is >= 16)
        return True
    if num("NECAT_RC_CARRY", 1) == 0 or num("NECAT_RC_RAGGED", 1) == 0 or num("NECAT_RC_WW", 1) == 0 or num("NECAT_FAST", 1) != 1 or num("NECAT_COOP_FILTER", 1) == 0:
        return True
    if (num("NECAT_RC_LISTB", 1) == 0 and not all_fused) or "NECAT_COOP_THRESHOLD" in env or num("NECAT_RC_MAXDIST", 1 << 20) < 300:
        return True
    return num("NECAT_SEED_WAVE", 1) == 0 or num("NECAT_ASM_LANE", 0) != 0 or num("NECAT_ASM_RC", 1) == 0


def load_library(path: Optional[str] = None, xcheck: bool = False) -> C.CDLL:
    """dlopen the in-tree library (xcheck: the tests' cross-check build of the same sources); never falls back to anything else."""
    global _lib, _lib_xcheck
    if xcheck:
        if _lib_xcheck is not None:
            return _lib_xcheck
    elif _lib is not None:
        return _lib
    if xcheck:
        p = path or _build.LIB_XCHECK
        if not os.path.exists(p):
            raise RuntimeError("libnecat_hip_xcheck.so is not built (%s): run `python -m necat_amd.build` (build_xcheck)" % p)
    else:
        p = path or os.environ.get("NECAT_HIP_LIB") or _build.LIB       # NECAT_HIP_LIB: an instrumented build of the same sources (tools/seed_prof.sh)
        if not os.path.exists(p):
            raise RuntimeError("libnecat_hip.so is not built (%s): run `python -m necat_amd.build`" % p)
    # (the host program's job since round 6 - the library no longer sets it when it is loaded; read by the HIP runtime at its first call)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    lib = C.CDLL(p)
    lib.necat_abi_version.restype = C.c_int
    if lib.necat_abi_version() != ABI_VERSION:
        raise RuntimeError("%s has ABI version %d, this binding was written for %d (include/necat_hip.h: NECAT_ABI_VERSION)" % (p, lib.necat_abi_version(), ABI_VERSION))
    vp, u64p, i32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
    lib.necat_get_timings_sized.argtypes = [vp, vp, C.c_size_t]
    lib.necat_get_shard_timings_sized.argtypes = [vp, vp, C.c_size_t]
    lib.necat_default_options.argtypes = [C.POINTER(MapOptions)]
    lib.necat_default_options.restype = None
    lib.necat_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.necat_ctx_destroy.argtypes = [vp]
    lib.necat_ctx_destroy.restype = None
    lib.necat_ctx_trim.argtypes = [vp]
    lib.necat_ctx_trim.restype = None
    lib.necat_pcan_partition.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int)]
    lib.necat_comm_selftest_rccl.argtypes = [vp, C.c_uint64]
    lib.necat_comm_selftest_rccl2.argtypes = [vp, C.c_uint64]
    lib.necat_last_error.argtypes = [vp]
    lib.necat_last_error.restype = C.c_char_p
    lib.necat_device_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.necat_volume_upload.argtypes = [vp, vp, C.c_uint64, vp, vp, C.c_uint64, C.POINTER(vp)]
    lib.necat_volume_free.argtypes = [vp, vp]
    lib.necat_volume_pack.argtypes = [vp, C.c_char_p, C.c_uint64, vp, vp, C.c_uint64, vp, C.POINTER(vp)]
    lib.necat_volume_free.restype = None
    lib.necat_index_build.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.necat_index_size.argtypes = [vp, u64p, u64p]
    lib.necat_index_download.argtypes = [vp, vp, vp, vp]
    lib.necat_index_free.argtypes = [vp, vp]
    lib.necat_index_sparse_size.argtypes = [vp, u64p, u64p]
    lib.necat_index_download_sparse.argtypes = [vp, vp, vp, vp, vp]
    lib.necat_index_free.restype = None
    lib.necat_find_candidates.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(MapOptions),
                                          C.POINTER(vp), u64p]
    lib.necat_extend.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_uint64, C.POINTER(MapOptions), C.c_int,
                                 C.POINTER(vp), u64p]
    lib.necat_map_pair.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(MapOptions), C.c_int,
                                   C.POINTER(vp), u64p, u64p]
    lib.necat_map_reference.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(MapOptions), C.POINTER(vp), u64p, u64p, u64p]
    lib.necat_onc_align_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_uint64, C.POINTER(MapOptions), C.c_int,
                                          C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.necat_asm_align_batch.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_uint64, C.c_double, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.necat_asm_plan_batch.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(vp), C.POINTER(vp)]
    lib.necat_gapped_strings.argtypes = [vp, C.c_uint64, vp, C.c_uint64, C.c_uint64, vp, C.c_uint64, C.c_uint64, vp, vp]
    lib.necat_cns_default_options.argtypes = [C.POINTER(CnsOptions)]
    lib.necat_cns_default_options.restype = None
    lib.necat_cns_load_partition.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u64p]
    lib.necat_cns_extension_batch.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.POINTER(CnsOptions),
                                              C.POINTER(C.POINTER(_CnsResult))]
    lib.necat_cns_result_free.argtypes = [C.POINTER(_CnsResult)]
    lib.necat_cns_result_free.restype = None
    lib.necat_edlib_align_batch.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, C.c_double,
                                            vp, vp, vp, C.POINTER(vp), C.POINTER(vp)]
    lib.necat_get_timings.argtypes = [vp, C.POINTER(Timings)]
    lib.necat_comm_create.argtypes = [vp, C.c_int, C.c_int, HOST_ALLGATHER_FN, vp, C.c_char_p, C.POINTER(vp)]
    lib.necat_comm_destroy.argtypes = [vp]
    lib.necat_comm_destroy.restype = None
    lib.necat_comm_transport.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.necat_get_shard_timings.argtypes = [vp, C.POINTER(ShardTimings)]
    lib.necat_index_build_sharded.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.necat_index_plan.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_double, C.POINTER(IndexPlan)]
    lib.necat_find_candidates_sharded.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(MapOptions), C.c_int, C.c_int,
                                                  C.POINTER(vp), u64p, u64p]
    lib.necat_map_pair_sharded.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(MapOptions), C.c_int, C.c_int, C.c_int,
                                           C.POINTER(vp), u64p, u64p, u64p]
    lib.necat_pair_chunk_reads.argtypes = [C.c_uint64, C.c_int]
    lib.necat_pair_schedule.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.necat_find_candidates_part.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(MapOptions), C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(vp), u64p]
    lib.necat_map_pair_part.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(MapOptions), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(vp), u64p, u64p]
    lib.necat_free.argtypes = [vp]
    lib.necat_free.restype = None
    for name in EXPORTED_SYMBOLS:
        getattr(lib, name)
    if xcheck:
        _lib_xcheck = lib
    else:
        _lib = lib
    return lib


def index_plan(nbases: int, k: int, nranks: int, link_gbs: float = 0.0) -> IndexPlan:
    """necat_index_plan: what a `nranks`-rank build of a volume's index costs either way (no device needed)"""
    p = IndexPlan()
    rc = load_library().necat_index_plan(nbases, k, nranks, link_gbs, C.byref(p))
    if rc:
        raise NecatError("necat_index_plan: %d" % rc)
    return p


def default_options(**kw) -> MapOptions:
    o = MapOptions()
    load_library().necat_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class NecatError(RuntimeError):
    pass


class Context:
    """necat_ctx + RAII wrappers of volumes and indexes."""

    def __init__(self, device: int = 0, xcheck: Optional[bool] = None):
        # xcheck None: the cross-check build exactly when the environment's knobs select one of its paths (needs_xcheck) - the product library otherwise
        self.xcheck = needs_xcheck() if xcheck is None else bool(xcheck)
        self.device = device
        self._xc, self._xc_key = None, None
        self.lib = load_library(xcheck=self.xcheck)
        h = C.c_void_p()
        rc = self.lib.necat_ctx_create(device, C.byref(h))
        if rc != 0:
            raise NecatError("necat_ctx_create(device=%d) failed with %d: no usable gfx950 GPU "
                             "(this library has no CPU fallback)" % (device, rc))
        self.h = h

    def close(self):
        if getattr(self, "_xc", None) is not None:
            self._xc.close()
            self._xc = None
        if getattr(self, "h", None):
            self.lib.necat_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise NecatError("%s failed (%d): %s" % (what, rc, self.lib.necat_last_error(self.h).decode()))

    def device_name(self) -> str:
        b = C.create_string_buffer(256)
        self.lib.necat_device_name(self.h, b, 256)
        return b.value.decode()

    def timings(self) -> Timings:
        t = Timings()
        self.lib.necat_get_timings(self.h, C.byref(t))
        return t

    # ---- one volume on several GPUs (include/necat_hip.h, "one reference volume on several GPUs")
    def shard_timings(self) -> ShardTimings:
        t = ShardTimings()
        self.lib.necat_get_shard_timings(self.h, C.byref(t))
        return t

    def comm(self, rank: int, nranks: int, allgather, transport: str = "auto") -> "Comm":
        """allgather(send: bytes) -> list of nranks bytes objects (rank order): the host-side exchange of the job's launcher
        (necat_amd.dist.torch_allgather for torch.distributed)"""
        return Comm(self, rank, nranks, allgather, transport)

    def build_index_sharded(self, comm: "Comm", ref: "Volume", k: int, max_occ: int) -> "Index":
        h = C.c_void_p()
        self._check(self.lib.necat_index_build_sharded(self.h, comm.h, ref.h, k, max_occ, C.byref(h)), "necat_index_build_sharded")
        return Index(self, h, k)

    def find_candidates_sharded(self, comm: "Comm", ix: "Index", ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int,
                                opt: MapOptions, pairwise: bool = True, chunk_reads: int = 64, root: int = 0):
        """(records - all ranks' on the root, this rank's elsewhere -, this rank's own count)"""
        p = C.c_void_p()
        n, nl = C.c_uint64(), C.c_uint64()
        self._check(self.lib.necat_find_candidates_sharded(self.h, comm.h, ix.h, ref.h, reads.h, read_start_id, ref_start_id, 1 if pairwise else 0,
                                                           C.byref(opt), chunk_reads, root, C.byref(p), C.byref(n), C.byref(nl)),
                    "necat_find_candidates_sharded")
        return self._take(p, n.value, CANDIDATE_DTYPE), int(nl.value)

    def map_pair_sharded(self, comm: "Comm", ix: "Index", ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int, opt: MapOptions,
                         pairwise: bool = True, tail_match_len: int = 1, chunk_reads: int = 64, root: int = 0):
        """(M4 records - all ranks' on the root -, this rank's own record count, this rank's candidates)"""
        p = C.c_void_p()
        n, nl, nc = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.necat_map_pair_sharded(self.h, comm.h, ix.h, ref.h, reads.h, read_start_id, ref_start_id, 1 if pairwise else 0,
                                                    C.byref(opt), tail_match_len, chunk_reads, root, C.byref(p), C.byref(n), C.byref(nl), C.byref(nc)),
                    "necat_map_pair_sharded")
        return self._take(p, n.value, M4_DTYPE), int(nl.value), int(nc.value)

    # ---- volumes
    def upload_volume(self, pac: np.ndarray, nbases: int, offsets: np.ndarray, sizes: np.ndarray) -> "Volume":
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        sz = np.ascontiguousarray(sizes, dtype=np.uint64)
        h = C.c_void_p()
        self._check(self.lib.necat_volume_upload(self.h, pac.ctypes.data, nbases, off.ctypes.data, sz.ctypes.data,
                                                 off.shape[0], C.byref(h)), "necat_volume_upload")
        return Volume(self, h, nbases, off.astype(np.int64), sz.astype(np.int64))

    def load_volume(self, path: str) -> "Volume":
        pac, off, sz, names = read_volume(path)
        v = self.upload_volume(pac, int(sz.sum()), off, sz)
        v.names = names
        return v

    def load_merged_volumes(self, wrk_dir: str) -> "Volume":
        """all volumes of a work directory as ONE read set with global ids (merge_volumes, common/makedb_aux.c:137-153) -
        what the consensus stage works on; .codes keeps the byte-coded bases for the callers' gapped strings"""
        from .synth import pack_2bit, unpack_2bit
        _, _, vols = load_volumes_info(wrk_dir)
        codes, sizes, names = [], [], []
        for path, _, _ in vols:
            pac, off, sz, nm = read_volume(path)
            codes.append(unpack_2bit(pac, int(sz.sum())))
            sizes.append(sz)
            names += nm
        codes = np.concatenate(codes) if codes else np.zeros(0, np.uint8)
        sizes = np.concatenate(sizes).astype(np.int64) if sizes else np.zeros(0, np.int64)
        off = np.zeros(sizes.shape[0], dtype=np.int64)
        if sizes.shape[0]:
            off[1:] = np.cumsum(sizes)[:-1]
        v = self.upload_volume(pack_2bit(codes), int(sizes.sum()), off, sizes)
        v.names, v.codes = names, codes
        return v

    def build_index(self, ref: "Volume", k: int, max_occ: int) -> "Index":
        h = C.c_void_p()
        self._check(self.lib.necat_index_build(self.h, ref.h, k, max_occ, C.byref(h)), "necat_index_build")
        return Index(self, h, k)

    def find_candidates(self, ix: "Index", ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int,
                        opt: MapOptions, pairwise: bool = True) -> np.ndarray:
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self.lib.necat_find_candidates(self.h, ix.h, ref.h, reads.h, read_start_id, ref_start_id,
                                                   1 if pairwise else 0, C.byref(opt), C.byref(p), C.byref(n)),
                    "necat_find_candidates")
        return self._take(p, n.value, CANDIDATE_DTYPE)

    def extend(self, ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int, cands: np.ndarray,
               opt: MapOptions, tail_match_len: int = 1) -> np.ndarray:
        cands = np.ascontiguousarray(cands, dtype=CANDIDATE_DTYPE)
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self.lib.necat_extend(self.h, ref.h, reads.h, read_start_id, ref_start_id, cands.ctypes.data,
                                          cands.shape[0], C.byref(opt), tail_match_len, C.byref(p), C.byref(n)),
                    "necat_extend")
        return self._take(p, n.value, M4_DTYPE)

    def map_pair(self, ix: "Index", ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int, opt: MapOptions,
                 pairwise: bool = True, tail_match_len: int = 1):
        """find_candidates + extend with the candidates kept on the device: (M4 records, number of candidates)."""
        p = C.c_void_p()
        n, nc = C.c_uint64(), C.c_uint64()
        self._check(self.lib.necat_map_pair(self.h, ix.h, ref.h, reads.h, read_start_id, ref_start_id, 1 if pairwise else 0,
                                            C.byref(opt), tail_match_len, C.byref(p), C.byref(n), C.byref(nc)), "necat_map_pair")
        return self._take(p, n.value, M4_DTYPE), int(nc.value)

    def find_candidates_part(self, ix: "Index", ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int, opt: MapOptions,
                             chunk_reads: int, slot_lo: int, slot_hi: int, slots: int, pairwise: bool = True) -> np.ndarray:
        """necat_find_candidates for the query chunks c with slot_lo <= c % slots < slot_hi (one unit of the pair scheduler)"""
        p = C.c_void_p()
        n = C.c_uint64()
        self._check(self.lib.necat_find_candidates_part(self.h, ix.h, ref.h, reads.h, read_start_id, ref_start_id, 1 if pairwise else 0, C.byref(opt),
                                                        chunk_reads, slot_lo, slot_hi, slots, C.byref(p), C.byref(n)), "necat_find_candidates_part")
        return self._take(p, n.value, CANDIDATE_DTYPE)

    def map_pair_part(self, ix: "Index", ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int, opt: MapOptions,
                      chunk_reads: int, slot_lo: int, slot_hi: int, slots: int, pairwise: bool = True, tail_match_len: int = 1):
        """necat_map_pair for one unit of the pair scheduler: (M4 records, number of candidates)"""
        p = C.c_void_p()
        n, nc = C.c_uint64(), C.c_uint64()
        self._check(self.lib.necat_map_pair_part(self.h, ix.h, ref.h, reads.h, read_start_id, ref_start_id, 1 if pairwise else 0, C.byref(opt), tail_match_len,
                                                 chunk_reads, slot_lo, slot_hi, slots, C.byref(p), C.byref(n), C.byref(nc)), "necat_map_pair_part")
        return self._take(p, n.value, M4_DTYPE), int(nc.value)

    def map_reference(self, ix: "Index", ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int, opt: MapOptions):
        """rm_search_one_volume for every read of `reads`: (M4 records, number of candidates, records from the rescue pair)."""
        p = C.c_void_p()
        n, nc, nr = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self.lib.necat_map_reference(self.h, ix.h, ref.h, reads.h, read_start_id, ref_start_id, C.byref(opt), C.byref(p), C.byref(n),
                                                 C.byref(nc), C.byref(nr)), "necat_map_reference")
        return self._take(p, n.value, M4_DTYPE), int(nc.value), int(nr.value)

    def onc_align_batch(self, ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int, cands: np.ndarray,
                        opt: MapOptions, tail_match_len: int = 4):
        """onc_align with the alignment itself for every candidate (the consensus stage's call):
        (alignments[ALIGNMENT_DTYPE], ops[uint8: 2 bits per column], ops_off[uint64, n + 1: byte offsets])."""
        cands = np.ascontiguousarray(cands, dtype=CANDIDATE_DTYPE)
        n = cands.shape[0]
        a, o, f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self.lib.necat_onc_align_batch(self.h, ref.h, reads.h, read_start_id, ref_start_id, cands.ctypes.data, n,
                                                   C.byref(opt), tail_match_len, C.byref(a), C.byref(o), C.byref(f)),
                    "necat_onc_align_batch")
        off = self._take(f, n + 1, np.dtype("<u8"))
        return self._take(a, n, ALIGNMENT_DTYPE), self._take(o, int(off[-1]), np.dtype("u1")), off

    def asm_align_batch(self, ref: "Volume", reads: "Volume", read_start_id: int, ref_start_id: int, anchors: np.ndarray, error: float = 0.5,
                        min_align_size: int = 400):
        """blockwise_edlib_align (oc2asmpm's block aligner: 2048-bp blocks, tail match length 8) for every anchor; returns as onc_align_batch"""
        anchors = np.ascontiguousarray(anchors, dtype=ASM_ANCHOR_DTYPE)
        n = anchors.shape[0]
        a, o, f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self.lib.necat_asm_align_batch(self.h, ref.h, reads.h, read_start_id, ref_start_id, anchors.ctypes.data, n, error, min_align_size,
                                                   C.byref(a), C.byref(o), C.byref(f)), "necat_asm_align_batch")
        off = self._take(f, n + 1, np.dtype("<u8"))
        return self._take(a, n, ALIGNMENT_DTYPE), self._take(o, max(8, int(off[-1])), np.dtype("u1")), off

    def cns_load_partition(self, reads: "Volume", packed: np.ndarray):
        """order and cut of one candidate partition as oc2cns does it: (cands, tmpl_off, n_all)"""
        packed = np.ascontiguousarray(packed).view(np.uint8).reshape(-1)
        n = packed.shape[0] // 28
        c, o, a, nt = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._check(self.lib.necat_cns_load_partition(self.h, reads.h, packed.ctypes.data, n, C.byref(c), C.byref(o), C.byref(a),
                                                      C.byref(nt)), "necat_cns_load_partition")
        off = self._take(o, nt.value + 1, np.dtype("<u8"))
        return self._take(c, int(off[-1]), CANDIDATE_DTYPE), off, self._take(a, nt.value, np.dtype("<u8"))

    def cns_extension_batch(self, reads: "Volume", cands: np.ndarray, tmpl_off: np.ndarray, n_all, opt: CnsOptions) -> "CnsResult":
        """the consensus stage's extension loop for all templates of the call (include/necat_hip.h)"""
        cands = np.ascontiguousarray(cands, dtype=CANDIDATE_DTYPE)
        tmpl_off = np.ascontiguousarray(tmpl_off, dtype=np.uint64)
        na = None if n_all is None else np.ascontiguousarray(n_all, dtype=np.uint64)
        r = C.POINTER(_CnsResult)()
        self._check(self.lib.necat_cns_extension_batch(self.h, reads.h, cands.ctypes.data, tmpl_off.ctypes.data,
                                                       None if na is None else na.ctypes.data, tmpl_off.shape[0] - 1,
                                                       C.byref(opt), C.byref(r)), "necat_cns_extension_batch")
        return CnsResult(self.lib, r)

    def edlib_align_batch(self, seqs: np.ndarray, q_off, q_len, t_off, t_len, error: float = 0.5, want_ops: bool = True):
        """necat_edlib_align_batch, the block-by-block hook of the parity tests: it exists in the cross-check build only, so a product context hands the call to a
        cross-check context of its own (made on first use, with the knobs of the environment at that moment)"""
        if not self.xcheck:
            # (a context reads its knobs when it is made: a cross-check context per knob environment, so that a test's NECAT_* settings choose the kernels they name)
            key = tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("NECAT_")))
            if self._xc is None or self._xc_key != key:
                if self._xc is not None:
                    self._xc.close()
                self._xc, self._xc_key = Context(self.device, xcheck=True), key
            return self._xc.edlib_align_batch(seqs, q_off, q_len, t_off, t_len, error, want_ops)
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
        q_len = np.ascontiguousarray(q_len, dtype=np.int32)
        t_len = np.ascontiguousarray(t_len, dtype=np.int32)
        n = q_off.shape[0]
        dist = np.zeros(n, dtype=np.int32)
        qend = np.zeros(n, dtype=np.int32)
        tend = np.zeros(n, dtype=np.int32)
        ops = C.c_void_p()
        ops_off = C.c_void_p()
        self._check(self.lib.necat_edlib_align_batch(self.h, seqs.ctypes.data, seqs.shape[0], q_off.ctypes.data,
                                                     q_len.ctypes.data, t_off.ctypes.data, t_len.ctypes.data, n, error,
                                                     dist.ctypes.data, qend.ctypes.data, tend.ctypes.data,
                                                     C.byref(ops) if want_ops else None,
                                                     C.byref(ops_off) if want_ops else None), "necat_edlib_align_batch")
        o = oo = None
        if want_ops and n:
            oo = self._take(ops_off, n + 1, np.dtype("<u8"))
            o = self._take(ops, int(oo[-1]), np.dtype("u1"))
        return dist, qend, tend, o, oo

    def _take(self, p: C.c_void_p, n: int, dtype: np.dtype) -> np.ndarray:
        """Wrap a library-malloc'ed result array without copying; necat_free runs when the array dies."""
        if not p.value:
            return np.zeros(0, dtype=dtype)
        if n == 0:
            self.lib.necat_free(p)
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * dtype.itemsize)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=n)
        weakref.finalize(buf, self.lib.necat_free, C.c_void_p(p.value))   # arr keeps buf alive through .base
        return arr


class Comm:
    """necat_comm: the rank-to-rank data path (RCCL or HIP IPC) + the caller's host all-gather as a C callback"""

    def __init__(self, ctx: Context, rank: int, nranks: int, allgather, transport: str = "auto"):
        self.ctx, self.rank, self.nranks = ctx, rank, nranks

        def cb(_user, send, recv, nbytes):
            try:
                parts = allgather(C.string_at(send, nbytes))
                if len(parts) != nranks or any(len(x) != nbytes for x in parts):
                    return 2
                C.memmove(recv, b"".join(parts), nbytes * nranks)
                return 0
            except Exception:      # an exception must not cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._cb = HOST_ALLGATHER_FN(cb)          # kept alive as long as the communicator
        h = C.c_void_p()
        ctx._check(ctx.lib.necat_comm_create(ctx.h, rank, nranks, self._cb, None, transport.encode(), C.byref(h)), "necat_comm_create")
        self.h = h

    def transport(self) -> str:
        b = C.create_string_buffer(16)
        self.ctx.lib.necat_comm_transport(self.h, b, 16)
        return b.value.decode()

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.necat_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CnsResult:
    """necat_cns_result: templates / overlaps / ranges are numpy views, valid until free()"""

    def __init__(self, lib, r):
        self.lib, self.r = lib, r
        c = r.contents
        self.templates = self._view(c.templates, c.n_templates, CNS_TEMPLATE_DTYPE)
        self.overlaps = self._view(c.overlaps, c.n_overlaps, CNS_OVERLAP_DTYPE)
        self.ranges = self._view(c.ranges, 2 * c.n_ranges, np.dtype("<i4")).reshape(-1, 2)
        self.n_aligned, self.n_used, self.n_rounds = c.n_aligned, c.n_used, c.n_rounds
        self.device_ms, self.host_ms = c.device_ms, c.host_ms
        self.n_rescue_tried, self.n_rescued, self.rescue_ms = c.n_rescue_tried, c.n_rescued, c.rescue_ms

    @staticmethod
    def _view(p, n, dtype):
        if not p or n == 0:
            return np.zeros(0, dtype=dtype)
        return np.frombuffer((C.c_char * (n * dtype.itemsize)).from_address(p), dtype=dtype, count=n)

    def ops(self, ov) -> np.ndarray:
        """the packed alignment columns (2 bits each) of one overlap record"""
        n = (int(ov["align_size"]) + 3) // 4
        base = self.r.contents.ops[int(ov["ops_block"])]
        return np.frombuffer((C.c_char * n).from_address(base + int(ov["ops_off"])), dtype=np.uint8, count=n) if n else np.zeros(0, np.uint8)

    def free(self):
        if self.r:
            self.templates = self.overlaps = self.ranges = None
            self.lib.necat_cns_result_free(self.r)
            self.r = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def cns_options(**kw) -> CnsOptions:
    o = CnsOptions()
    load_library().necat_cns_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


class Volume:
    def __init__(self, ctx: Context, h, nbases: int, offsets: np.ndarray, sizes: np.ndarray):
        self.ctx, self.h, self.nbases, self.offsets, self.sizes = ctx, h, nbases, offsets, sizes
        self.names: List[str] = []

    @property
    def nseq(self) -> int:
        return int(self.sizes.shape[0])

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.necat_volume_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Index:
    def __init__(self, ctx: Context, h, k: int):
        self.ctx, self.h, self.k = ctx, h, k

    def sizes(self) -> Tuple[int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        self.ctx.lib.necat_index_size(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def download(self, want_stats: bool = True) -> Tuple[Optional[np.ndarray], np.ndarray]:
        T, n = self.sizes()
        stats = np.empty(T, dtype=np.uint64) if want_stats else None
        offs = np.empty(n, dtype=np.uint64)
        self.ctx._check(self.ctx.lib.necat_index_download(self.ctx.h, self.h, stats.ctypes.data if want_stats else None,
                                                          offs.ctypes.data if n else None), "necat_index_download")
        return stats, offs

    def sparse_sizes(self) -> Tuple[int, int]:
        """(pairs of (bits, base) words, non-zero table entries) of an index in the sparse layout; (0, 0) for a dense one"""
        a, b = C.c_uint64(), C.c_uint64()
        self.ctx.lib.necat_index_sparse_size(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def download_sparse(self):
        """(bits[T / 64], base[T / 64], compact, offset_list) of an index held in the sparse layout (IndexView, dev_common.h); None for a dense one"""
        npairs, ncomp = C.c_uint64(), C.c_uint64()
        self.ctx._check(self.ctx.lib.necat_index_sparse_size(self.h, C.byref(npairs), C.byref(ncomp)), "necat_index_sparse_size")
        if npairs.value == 0:
            return None
        _, n_off = self.sizes()
        pairs = np.empty(2 * npairs.value, dtype=np.uint64)
        comp = np.empty(max(1, ncomp.value), dtype=np.uint64)
        offs = np.empty(max(1, n_off), dtype=np.uint64)
        self.ctx._check(self.ctx.lib.necat_index_download_sparse(self.ctx.h, self.h, pairs.ctypes.data, comp.ctypes.data, offs.ctypes.data), "necat_index_download_sparse")
        return pairs[0::2].copy(), pairs[1::2].copy(), comp[:ncomp.value], offs[:n_off]

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.necat_index_free(self.ctx.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# --------------------------------------------------------------------------------------------------
# Python mirror of pm_main (pm_one_volume/pm_worker.c:338-400) for tests / bench: volume `vid`
# against every volume >= vid, records returned instead of written.
# --------------------------------------------------------------------------------------------------

def load_volumes_info(wrk_dir: str):
    """common/makedb_aux.c:78-118"""
    base = wrk_dir if wrk_dir.endswith("/") else wrk_dir + "/"
    with open(base + "reads_info.txt") as f:
        nv, nr = [int(x) for x in f.read().split()[:2]]
    vols = []
    with open(base + "volume_names.txt") as f:
        for _ in range(nv):
            parts = f.readline().split()
            vols.append((parts[0], int(parts[1]), int(parts[2])))
    return nv, nr, vols


def pack_columns(ops: np.ndarray) -> np.ndarray:
    """one op code (0..3) per column -> the library's packed form: 2 bits per column, 4 per byte, low bits first"""
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    n = ops.shape[0]
    pad = np.zeros(((n + 3) // 4) * 4, dtype=np.uint8)
    pad[:n] = ops & 3
    q = pad.reshape(-1, 4)
    return (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)


def unpack_columns(packed: np.ndarray, n: int) -> np.ndarray:
    p = np.ascontiguousarray(packed, dtype=np.uint8)[:(n + 3) // 4]
    return np.stack([p & 3, (p >> 2) & 3, (p >> 4) & 3, (p >> 6) & 3], axis=1).reshape(-1)[:n].astype(np.uint8)


def gapped_strings(ops: np.ndarray, n: int, qseq: np.ndarray, qoff: int, tseq: np.ndarray, toff: int):
    """necat_gapped_strings: `n` packed columns (2 bits each) -> (query_align, target_align) as bytes ("ACGT-")."""
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    if ops.shape[0] * 4 < n:
        raise ValueError("%d columns need %d bytes, got %d" % (n, (n + 3) // 4, ops.shape[0]))
    q = np.ascontiguousarray(qseq, dtype=np.uint8)
    t = np.ascontiguousarray(tseq, dtype=np.uint8)
    qa = C.create_string_buffer(max(1, n))
    ta = C.create_string_buffer(max(1, n))
    rc = load_library().necat_gapped_strings(ops.ctypes.data, n, q.ctypes.data, q.shape[0], qoff, t.ctypes.data, t.shape[0], toff, qa, ta)
    if rc != 0:
        raise NecatError("necat_gapped_strings failed with %d" % rc)
    return qa.raw[:n], ta.raw[:n]


def pcan_single_partition(packed: bytes) -> bytes:
    """oc2pcan (partition_candidates/pcan.c:39-103) when all reads fall in ONE partition: every record is kept
    and followed by its role-swapped twin (change_pcan_roles, common/gapped_candidate.c:54-69: the subject
    becomes the query, strands swap with them).  Record order inside a partition file is free."""
    a = np.frombuffer(packed, dtype="<u4").reshape(-1, 7)
    b = np.empty_like(a)
    w0 = a[:, 0]
    b[:, 0] = (w0 & np.uint32((1 << 30) - 1)) | ((w0 >> 31) << 30) | (((w0 >> 30) & 1) << 31)
    b[:, 1:4] = a[:, 4:7]
    b[:, 4:7] = a[:, 1:4]
    return np.concatenate([a, b]).tobytes()


def pack_candidates(c: np.ndarray) -> np.ndarray:
    """pack_candidate (common/gapped_candidate.c:13-30): 7 x u32 records."""
    out = np.zeros((c.shape[0], 7), dtype=np.uint32)
    item0 = np.minimum(c["score"], 1000000).astype(np.uint32)
    item0 |= (c["sdir"] == 1).astype(np.uint32) << 31
    item0 |= (c["qdir"] == 1).astype(np.uint32) << 30
    item0 |= (c["qoff"] == c["qbeg"]).astype(np.uint32) << 29
    out[:, 0] = item0
    out[:, 1] = c["sid"].astype(np.uint32)
    out[:, 2] = c["sbeg"].astype(np.uint32)
    out[:, 3] = c["send"].astype(np.uint32)
    out[:, 4] = c["qid"].astype(np.uint32)
    out[:, 5] = c["qbeg"].astype(np.uint32)
    out[:, 6] = c["qend"].astype(np.uint32)
    return out


def m4_text_lines(m: np.ndarray) -> List[bytes]:
    """DUMP_ASM_M4 (common/m4_record.h:72-97), numeric ids."""
    return [b"%d\t%d\t%.2f\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n" %
            (r["qid"], r["sid"], r["ident_perc"], r["vscore"], r["qdir"], r["qoff"], r["qend"], r["qsize"],
             r["sdir"], r["soff"], r["send"], r["ssize"]) for r in m]


def candidate_text_lines(c: np.ndarray) -> List[bytes]:
    """DUMP_GAPPED_CANDIDATE (common/gapped_candidate.h:26-42)."""
    return [b"%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n" %
            (r["qid"], r["sid"], r["score"], r["qdir"], r["qbeg"], r["qend"], r["qoff"], r["qsize"], r["sdir"],
             r["sbeg"], r["send"], r["soff"], r["ssize"]) for r in c]


PAIR_UNIT_DTYPE = np.dtype([("ref_vol", "<i4"), ("query_vol", "<i4"), ("slot_lo", "<i4"), ("slot_hi", "<i4")])


def pair_chunk_reads(query_reads: int, slots: int = 64) -> int:
    return int(load_library().necat_pair_chunk_reads(int(query_reads), slots))


def pair_schedule(vol_bases, nranks: int, slots: int = 64):
    """necat_pair_schedule (host arithmetic only, no device): (units[PAIR_UNIT_DTYPE], rank_off[nranks + 1], team[V, 2]) - rank g
    owns units[rank_off[g]:rank_off[g + 1]]; ranks team[v, 0] .. team[v, 1] (inclusive) work on reference volume v."""
    lib = load_library()
    vb = np.ascontiguousarray(vol_bases, dtype=np.uint64)
    V = int(vb.shape[0])
    pu, po, pt = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib.necat_pair_schedule(vb.ctypes.data, V, nranks, slots, C.byref(pu), C.byref(po), C.byref(pt))
    if rc:
        raise NecatError("necat_pair_schedule failed (%d)" % rc)
    off = np.frombuffer((C.c_char * (8 * (nranks + 1))).from_address(po.value), dtype=np.uint64).copy()
    n = int(off[nranks])
    units = np.frombuffer((C.c_char * (16 * max(n, 1))).from_address(pu.value), dtype=PAIR_UNIT_DTYPE)[:n].copy()
    team = np.frombuffer((C.c_char * (8 * V)).from_address(pt.value), dtype=np.int32).reshape(V, 2).copy()
    for q in (pu, po, pt):
        lib.necat_free(q)
    return units, off.astype(np.int64), team


def pm_main(ctx: Context, opt: MapOptions, vid: int, wrk_dir: str):
    """Returns (candidates, m4) over all volume pairs (vid, i >= vid); m4 is None for job 0."""
    nv, _, vols = load_volumes_info(wrk_dir)
    ref = ctx.load_volume(vols[vid][0])
    ix = ctx.build_index(ref, opt.kmer_size, opt.kmer_cnt_cutoff)
    cands, m4s = [], []
    for i in range(vid, nv):
        reads = ref if i == vid else ctx.load_volume(vols[i][0])
        c = ctx.find_candidates(ix, ref, reads, vols[i][1], vols[vid][1], opt, True)
        cands.append(c)
        if opt.job == 1:
            m4s.append(ctx.extend(ref, reads, vols[i][1], vols[vid][1], c, opt, 1))
        if reads is not ref:
            reads.free()
    ix.free()
    ref.free()
    return np.concatenate(cands), (np.concatenate(m4s) if opt.job == 1 else None)
