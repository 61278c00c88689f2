"""ctypes binding of ``libmonica_amd.so`` (C-ABI: ``include/monica_amd.h``).

The library holds the hand-written gfx950 kernels; there is no CPU execution path.  Loading
fails loudly when the shared object is missing -- build it with ``python -c 'import
__graft_entry__ as g; g.build()'`` or ``make -C monica_amd/csrc``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The host passes of the library (FASTQ parse, routing, index build) are OpenMP teams started from several Python threads
# at once; with an ACTIVE wait policy their idle threads spin on every core and the pipeline of aligner.py runs 15x slower
# (measured: 2.7 s instead of 0.16 s per GB of FASTQ).  Only a default: a policy the user has set is left alone.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
LIB_PATH = os.environ.get("MONICA_AMD_LIB") or os.path.join(_HERE, "libmonica_amd.so")   # (the override: an experimental build of the same ABI)

OK = 0
ERR_ARG, ERR_IO, ERR_FORMAT, ERR_NOMEM, ERR_HIP, ERR_NODEVICE, ERR_UNSUPPORTED, ERR_RANGE = range(-1, -9, -1)
UNMAPPED = -1
AMBIGUOUS = -2
SKIPPED = -3            # MNC_SKIPPED: a read outside the kernels' limits (include/monica_amd.h, "limits"): not classified, no hits

HIT_DTYPE = np.dtype([("rid", "<i4"), ("mapq", "<i4"), ("nm", "<i4"), ("mlen", "<i4")])
REG_DTYPE = np.dtype([("id", "<i4"), ("parent", "<i4"), ("rid", "<i4"), ("rev", "<i4"),
                      ("rs", "<i4"), ("re", "<i4"), ("qs", "<i4"), ("qe", "<i4"),
                      ("score", "<i4"), ("score0", "<i4"), ("cnt", "<i4"), ("as", "<i4"),
                      ("mlen", "<i4"), ("blen", "<i4"), ("subsc", "<i4"), ("n_sub", "<i4"),
                      ("mapq", "<i4"), ("hash", "<u4"),
                      ("dp_score", "<i4"), ("dp_max", "<i4"), ("dp_max2", "<i4"), ("n_ambi", "<i4"),
                      ("n_cigar", "<i4"), ("flags", "<i4")])
MZ_DTYPE = np.dtype([("hash", "<u4"), ("pos_strand", "<u4")])
ANCHOR_DTYPE = np.dtype([("x", "<u8"), ("y", "<u8")])

N_STAGES = 22
(STAGE_PACK, STAGE_SKETCH, STAGE_PARTITION, STAGE_PROBE, STAGE_COLLECT, STAGE_SORT, STAGE_SORT2, STAGE_CHAIN,
 STAGE_BACKTRACK, STAGE_REGIONS, STAGE_GATHER, STAGE_DP_PLAN, STAGE_DP_ALIGN, STAGE_DP_STITCH,
 STAGE_DP_POST, STAGE_DP_FILL, STAGE_DP_FILL_T1, STAGE_DP_FILL_T2, STAGE_DP_FILL_T3, STAGE_DP_EXT, STAGE_DP_FILL_TM, STAGE_DP_LFILL) = range(N_STAGES)
(DUMP_MINIMIZERS, DUMP_MZ_OFFSETS, DUMP_ANCHORS, DUMP_AN_OFFSETS, DUMP_CHAIN_F, DUMP_CHAIN_P,
 DUMP_CHAIN_V, DUMP_REGS, DUMP_REG_OFFSETS, DUMP_REP_LEN, DUMP_CIGARS, DUMP_SEGS) = range(1, 13)
SEG_DTYPE = np.dtype([(k, np.int32) for k in ("read", "reg", "kind", "rid", "rev", "ts", "tlen", "qs", "qlen", "w", "zdrop", "flag",
                                              "ai", "big", "n_cigar", "zdropped", "zdrop_code", "max", "max_t", "max_q", "score",
                                              "reach_end", "mqe_t", "pad")] + [("cig_off", np.int64)])
CONTRACT_DP, CONTRACT_CHAIN = 0, 1

# every symbol include/monica_amd.h declares (checked by tests/test_capi.py)
EXPORTS = [
    "mnc_strerror", "mnc_last_error", "mnc_device_count", "mnc_device_name", "mnc_device_mem_info",
    "mnc_index_build", "mnc_index_build_mem", "mnc_index_build_device", "mnc_index_build_mem_device", "mnc_index_save", "mnc_index_save_mmi", "mnc_index_load", "mnc_index_free",
    "mnc_index_info", "mnc_index_contig_name", "mnc_index_contig_len", "mnc_index_contig_genome",
    "mnc_index_genome_name", "mnc_index_genome_len", "mnc_index_dump", "mnc_index_set_mid_occ",
    "mnc_engine_create", "mnc_engine_destroy", "mnc_engine_stream", "mnc_engine_device_bytes", "mnc_index_device_bytes", "mnc_engine_set_index",
    "mnc_classify_batch", "mnc_engine_prefetch", "mnc_engine_prefetch_cancel", "mnc_classify_device", "mnc_engine_sync", "mnc_engine_fetch_hits",
    "mnc_counts", "mnc_best_hit",
    "mnc_engine_set_profiling", "mnc_engine_set_debug", "mnc_engine_set_contract", "mnc_index_set_host_tables", "mnc_index_set_region_bits", "mnc_engine_dump_tables",
    "mnc_comm_unique_id", "mnc_comm_init_rank", "mnc_comm_destroy", "mnc_comm_count", "mnc_allreduce_counts", "mnc_allgather_summaries", "mnc_shard_summary", "mnc_merge_summaries", "mnc_engine_get_timings", "mnc_stage_name", "mnc_stage_kernel",
    "mnc_engine_get_counters", "mnc_engine_dump",
    "mnc_fastq_open", "mnc_fastq_close", "mnc_fastq_next", "mnc_fastq_detach_batch", "mnc_fastq_remaining", "mnc_fastq_bases", "mnc_fastq_offsets",
    "mnc_fastq_quals", "mnc_fastq_title", "mnc_fastq_route",
    "mnc_hitmap_create", "mnc_hitmap_load", "mnc_hitmap_save", "mnc_hitmap_free", "mnc_hitmap_size",
    "mnc_hitmap_update", "mnc_hitmap_n_names", "mnc_hitmap_name", "mnc_host_alloc", "mnc_host_free", "mnc_host_set_io_workers",
    "mnc_synth_genome", "mnc_synth_diverge", "mnc_synth_reads", "mnc_synth_reads_device", "mnc_version",
]


class IndexInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("w", C.c_int32), ("n_contigs", C.c_int32), ("n_genomes", C.c_int32),
                ("mid_occ", C.c_int32), ("reserved", C.c_int32), ("n_keys", C.c_int64),
                ("n_occ", C.c_int64), ("total_len", C.c_int64), ("table_slots", C.c_int64),
                ("device_bytes", C.c_int64)]


class MncError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = lib().mnc_strerror(code).decode()
        super().__init__(f"{msg}" + (f": {detail}" if detail else ""))


_LIB = None


def lib():
    """Load the shared object once.  Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                          "(run __graft_entry__.build() or make -C monica_amd/csrc)")
    # PyTorch wheels bundle their own libamdhip64 under the same SONAME.  Two HIP runtimes in
    # one process cannot both open the GPU, so when torch is installed let it load its copy
    # first; our NEEDED entry then resolves to that already-loaded runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, u32, cp = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32, C.c_char_p
    pp = C.POINTER(vp)

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype, f.argtypes = res, args

    sig("mnc_strerror", cp, [i32])
    sig("mnc_last_error", cp, [])
    sig("mnc_version", cp, [])
    sig("mnc_device_count", i32, [C.POINTER(i32)])
    sig("mnc_device_name", i32, [i32, cp, C.c_size_t])
    sig("mnc_device_mem_info", i32, [i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
    sig("mnc_index_build", i32, [cp, cp, i32, i32, pp])
    sig("mnc_index_build_mem", i32, [i32, C.POINTER(cp), C.POINTER(cp), C.POINTER(i64), i32, i32, pp])
    sig("mnc_index_build_device", i32, [cp, cp, i32, i32, i32, pp])
    sig("mnc_index_build_mem_device", i32, [i32, C.POINTER(cp), C.POINTER(cp), C.POINTER(i64), i32, i32, i32, pp])
    sig("mnc_index_save", i32, [vp, cp])
    sig("mnc_index_save_mmi", i32, [vp, cp])
    sig("mnc_index_load", i32, [cp, pp])
    sig("mnc_index_free", None, [vp])
    sig("mnc_index_info", i32, [vp, C.POINTER(IndexInfo)])
    sig("mnc_index_contig_name", cp, [vp, i32])
    sig("mnc_index_contig_len", i64, [vp, i32])
    sig("mnc_index_contig_genome", i32, [vp, i32])
    sig("mnc_index_genome_name", cp, [vp, i32])
    sig("mnc_index_genome_len", i64, [vp, i32])
    sig("mnc_index_dump", i32, [vp, vp, vp, i64, C.POINTER(i64)])
    sig("mnc_index_set_mid_occ", i32, [vp, i32])
    sig("mnc_engine_create", i32, [vp, i32, pp])
    sig("mnc_engine_destroy", None, [vp])
    sig("mnc_engine_stream", vp, [vp])
    sig("mnc_engine_prefetch", i32, [vp, vp, vp, u32, C.POINTER(C.c_int)])
    sig("mnc_engine_prefetch_cancel", i32, [vp])
    sig("mnc_engine_set_index", i32, [vp, vp])
    sig("mnc_engine_device_bytes", i32, [vp, C.POINTER(C.c_int64)])
    sig("mnc_index_device_bytes", i32, [vp, i32, C.POINTER(C.c_int64)])
    sig("mnc_classify_batch", i32, [vp, vp, vp, u32, i32, vp, vp, vp])
    sig("mnc_classify_device", i32, [vp, vp, vp, u32, i64, i32, i32, vp, vp, vp, vp])
    sig("mnc_engine_sync", i32, [vp])
    sig("mnc_engine_fetch_hits", i32, [vp, vp, vp, i64, C.POINTER(i64)])
    sig("mnc_counts", i32, [vp, vp, vp, vp, u32, i32, vp])
    sig("mnc_best_hit", i32, [vp, i32, C.POINTER(i32)])
    sig("mnc_engine_set_profiling", i32, [vp, i32])
    sig("mnc_engine_set_debug", i32, [vp, i32])
    sig("mnc_index_set_host_tables", i32, [vp, i32])
    sig("mnc_index_set_region_bits", i32, [vp, i32])
    sig("mnc_engine_dump_tables", i32, [vp, vp, C.c_int64, vp])
    sig("mnc_engine_set_contract", i32, [vp, i32])
    sig("mnc_comm_unique_id", i32, [vp])
    sig("mnc_comm_init_rank", i32, [vp, i32, i32, C.POINTER(vp)])
    sig("mnc_comm_destroy", i32, [vp])
    sig("mnc_comm_count", i32, [vp, C.POINTER(i32)])
    sig("mnc_allreduce_counts", i32, [vp, i32, vp, vp])
    sig("mnc_allgather_summaries", i32, [vp, vp, C.c_size_t, vp, vp])
    sig("mnc_shard_summary", i32, [vp, vp, vp, i64, i32, vp, vp])
    sig("mnc_merge_summaries", i32, [vp, i32, i64, vp, vp, vp, vp, vp])
    sig("mnc_engine_get_timings", i32, [vp, vp, vp, i32])
    sig("mnc_stage_name", cp, [i32])
    sig("mnc_stage_kernel", cp, [i32])
    sig("mnc_engine_get_counters", i32, [vp, vp, i32])
    sig("mnc_engine_dump", i32, [vp, i32, vp, i64, C.POINTER(i64)])
    sig("mnc_synth_genome", i32, [u64, i64, vp])
    sig("mnc_synth_diverge", i32, [vp, i64, u64, i32, vp])
    sig("mnc_synth_reads", i32, [i32, C.POINTER(vp), C.POINTER(i64), u64, i64, i32, i32,
                                 i32, i32, i32, i32, vp, vp])
    sig("mnc_synth_reads_device", i32, [i32, vp, vp, u64, i64, i32, i32, i32, i32, i32, i32, vp, vp, vp])
    sig("mnc_fastq_open", i32, [cp, pp])
    sig("mnc_fastq_close", None, [vp])
    sig("mnc_fastq_detach_batch", i32, [vp, pp])
    sig("mnc_fastq_remaining", i32, [vp, C.POINTER(i64)])
    sig("mnc_fastq_next", i32, [vp, u32, u64, C.POINTER(u32)])
    sig("mnc_fastq_bases", vp, [vp])
    sig("mnc_fastq_offsets", vp, [vp])
    sig("mnc_fastq_quals", vp, [vp])
    sig("mnc_fastq_title", i32, [vp, u32, C.POINTER(vp), C.POINTER(u32), C.POINTER(u32)])
    sig("mnc_fastq_route", i32, [vp, vp, vp, C.POINTER(cp), i32, C.POINTER(cp)])
    sig("mnc_hitmap_create", i32, [pp])
    sig("mnc_hitmap_load", i32, [cp, pp])
    sig("mnc_hitmap_save", i32, [vp, cp])
    sig("mnc_hitmap_free", None, [vp])
    sig("mnc_hitmap_size", i64, [vp])
    sig("mnc_hitmap_update", i32, [vp, vp, vp, vp, vp, vp, vp])
    sig("mnc_hitmap_n_names", i32, [vp])
    sig("mnc_hitmap_name", cp, [vp, i32])
    sig("mnc_host_alloc", vp, [C.c_size_t])
    sig("mnc_host_free", None, [vp])
    sig("mnc_host_set_io_workers", i32, [i32])
    _LIB = L
    return L


def check(code):
    if code != OK:
        raise MncError(code, lib().mnc_last_error().decode(errors="replace"))


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else str(s).encode()


class Comm:
    """An RCCL communicator made through the C-ABI (`mnc_comm_*`): rank 0 calls `Comm.unique_id()`,
    hands the 128 bytes to the other ranks by any means, every rank constructs `Comm(id, n, rank)`
    with its device current."""

    def __init__(self, unique_id, n_ranks, rank):
        h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        check(lib().mnc_comm_init_rank(buf, int(n_ranks), int(rank), C.byref(h)))
        self._h, self.n_ranks, self.rank = h, int(n_ranks), int(rank)

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        check(lib().mnc_comm_unique_id(buf))
        return bytes(buf)

    def count(self):
        """ncclCommCount: the number of ranks RCCL itself reports for this communicator."""
        n = C.c_int32(0)
        check(lib().mnc_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def allreduce_counts(self, d_counts_ptr, n, stream=None):
        check(lib().mnc_allreduce_counts(C.c_void_p(d_counts_ptr), int(n), self._h, C.c_void_p(stream or 0)))

    def allgather_summaries(self, d_send_ptr, d_recv_ptr, bytes_per_rank, stream=None):
        check(lib().mnc_allgather_summaries(C.c_void_p(d_send_ptr), C.c_void_p(d_recv_ptr), int(bytes_per_rank), self._h,
                                            C.c_void_p(stream or 0)))

    def close(self):
        if getattr(self, "_h", None):
            lib().mnc_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_summary_device(d_assign, d_best, d_nhits, n, rid_offset, d_out, stream=None):
    """`mnc_shard_summary`: one index part's 20-byte per-read records [n][5] from the device outputs of
    `mnc_classify_device`; all arguments but `n` / `rid_offset` are device addresses."""
    check(lib().mnc_shard_summary(C.c_void_p(d_assign), C.c_void_p(d_best), C.c_void_p(d_nhits), int(n), int(rid_offset),
                                  C.c_void_p(d_out), C.c_void_p(stream or 0)))


def merge_summaries_device(d_parts, n_parts, n, d_assign, d_nm=0, d_mlen=0, d_total=0, stream=None):
    """`mnc_merge_summaries`: best_hit over [n_parts][n][5] records in part order (device addresses)."""
    check(lib().mnc_merge_summaries(C.c_void_p(d_parts), int(n_parts), int(n), C.c_void_p(d_assign), C.c_void_p(d_nm or 0),
                                    C.c_void_p(d_mlen or 0), C.c_void_p(d_total or 0), C.c_void_p(stream or 0)))


def set_io_workers(n):
    """`mnc_host_set_io_workers`: the FASTQ readers' parse / routing teams share the cores over `n` samples in flight."""
    check(lib().mnc_host_set_io_workers(int(max(1, n))))


def pinned_array(a):
    """A copy of `a` in page-locked host memory (mnc_host_alloc), as the FASTQ reader's batches are."""
    a = np.ascontiguousarray(a)
    L = lib()
    L.mnc_host_alloc.restype = C.c_void_p
    L.mnc_host_alloc.argtypes = [C.c_size_t]
    p = L.mnc_host_alloc(max(a.nbytes, 1))
    out = np.frombuffer((C.c_uint8 * max(a.nbytes, 1)).from_address(p), dtype=np.uint8)[:a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out                                          # (freed with the process)


def device_count():
    n = C.c_int(0)
    check(lib().mnc_device_count(C.byref(n)))
    return n.value


def device_mem_info(device=0):
    """(free, total) HBM bytes of a device."""
    f, t = C.c_int64(0), C.c_int64(0)
    check(lib().mnc_device_mem_info(device, C.byref(f), C.byref(t)))
    return f.value, t.value


# ---------------------------------------------------------------------------------- index
class Index:
    """Owner of an ``mnc_index`` handle (the object mappy.Aligner is for monica)."""

    def __init__(self, handle):
        self._h = handle
        info = self.info()
        self.k, self.w = info.k, info.w
        L = lib()
        self.contig_names = [L.mnc_index_contig_name(handle, i).decode() for i in range(info.n_contigs)]
        self.contig_lens = [int(L.mnc_index_contig_len(handle, i)) for i in range(info.n_contigs)]
        self.contig_genome = np.array([L.mnc_index_contig_genome(handle, i) for i in range(info.n_contigs)],
                                      dtype=np.int32)
        self.genome_names = [L.mnc_index_genome_name(handle, g).decode() for g in range(info.n_genomes)]
        self.genome_lens = [int(L.mnc_index_genome_len(handle, g)) for g in range(info.n_genomes)]

    @classmethod
    def build(cls, fasta_path, out_path=None, k=15, w=10, device=None):
        h = C.c_void_p()
        if device is None:
            check(lib().mnc_index_build(_b(fasta_path), _b(out_path) if out_path else None, k, w, C.byref(h)))
        else:                                       # sketch + sort on that device: the same index
            check(lib().mnc_index_build_device(_b(fasta_path), _b(out_path) if out_path else None, k, w, int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def from_seqs(cls, names, seqs, k=15, w=10, device=None):
        n = len(names)
        bn = [_b(x) for x in names]
        bs = [x if isinstance(x, (bytes, bytearray)) else (x.tobytes() if isinstance(x, np.ndarray) else _b(x))
              for x in seqs]
        an = (C.c_char_p * n)(*bn)
        as_ = (C.c_char_p * n)(*bs)
        al = (C.c_int64 * n)(*[len(x) for x in bs])
        h = C.c_void_p()
        if device is None:
            check(lib().mnc_index_build_mem(n, an, as_, al, k, w, C.byref(h)))
        else:                                       # sketch + sort on that device: the same index
            check(lib().mnc_index_build_mem_device(n, an, as_, al, k, w, int(device), C.byref(h)))
        return cls(h)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        check(lib().mnc_index_load(_b(path), C.byref(h)))
        return cls(h)

    def save(self, path, mmi=False):
        """This library's own file (loads without a sort), or -- `mmi=True` -- minimap2's format, the one mappy writes at
        aligner.py:45-46; `load` reads either."""
        check((lib().mnc_index_save_mmi if mmi else lib().mnc_index_save)(self._h, _b(path)))

    def info(self):
        info = IndexInfo()
        check(lib().mnc_index_info(self._h, C.byref(info)))
        return info

    def device_bytes(self, device):
        """HBM this index holds on one device (info().device_bytes sums over all devices)."""
        n = C.c_int64(0)
        check(lib().mnc_index_device_bytes(self._h, int(device), C.byref(n)))
        return int(n.value)

    @property
    def mid_occ(self):
        return self.info().mid_occ

    def set_mid_occ(self, v):
        check(lib().mnc_index_set_mid_occ(self._h, int(v)))

    def dump(self):
        n = C.c_int64(0)
        lib().mnc_index_dump(self._h, None, None, 0, C.byref(n))
        h = np.zeros(n.value, dtype=np.uint64)
        y = np.zeros(n.value, dtype=np.uint64)
        check(lib().mnc_index_dump(self._h, h.ctypes.data, y.ctypes.data, n.value, C.byref(n)))
        return h, y

    def close(self):
        if getattr(self, "_h", None):
            lib().mnc_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------- engine
class Engine:
    """One HIP stream + HBM workspace on one device, bound to one index."""

    def __init__(self, index, device=0):
        self.index = index
        self.device = device
        h = C.c_void_p()
        check(lib().mnc_engine_create(index._h, device, C.byref(h)))
        self._h = h
        self.n_reads = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().mnc_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return lib().mnc_engine_stream(self._h)

    def set_index(self, index):
        """Another index part behind the same stream and batch buffers (aligner.py:91-103 rebinds `index`)."""
        check(lib().mnc_engine_set_index(self._h, index._h))
        self.index = index

    def device_bytes(self):
        """HBM held by this engine's own batch buffers."""
        n = C.c_int64(0)
        check(lib().mnc_engine_device_bytes(self._h, C.byref(n)))
        return n.value

    def classify(self, bases, offsets, min_mapq=60):
        """Host buffers in, host arrays out: (assign, best, nhits)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        assign = np.empty(n, dtype=np.int32)
        best = np.zeros(n, dtype=HIT_DTYPE)
        nhits = np.zeros(n, dtype=np.int32)
        check(lib().mnc_classify_batch(self._h, bases.ctypes.data, offsets.ctypes.data, n, min_mapq,
                                       assign.ctypes.data, best.ctypes.data, nhits.ctypes.data))
        self.n_reads = n
        return assign, best, nhits

    def prefetch_ptr(self, bases_ptr, offsets_ptr, n):
        """Start the H2D copy of the NEXT batch (same pointers as its classify_ptr call will pass) behind the kernels
        of the one being classified.  False: a prefetched batch is still waiting for its call; nothing was done."""
        started = C.c_int(0)
        check(lib().mnc_engine_prefetch(self._h, bases_ptr, offsets_ptr, n, C.byref(started)))
        return bool(started.value)

    def prefetch_cancel(self):
        """Forget an announced batch that will not be classified (its host arrays may be reused afterwards)."""
        check(lib().mnc_engine_prefetch_cancel(self._h))

    def classify_ptr(self, bases_ptr, offsets_ptr, n, min_mapq=60):
        """As classify(), on caller-owned host buffers given by address (e.g. a FastqReader batch)."""
        assign = np.empty(n, dtype=np.int32)
        best = np.zeros(n, dtype=HIT_DTYPE)
        nhits = np.zeros(n, dtype=np.int32)
        check(lib().mnc_classify_batch(self._h, bases_ptr, offsets_ptr, n, min_mapq,
                                       assign.ctypes.data, best.ctypes.data, nhits.ctypes.data))
        self.n_reads = n
        return assign, best, nhits

    def classify_device(self, d_bases, d_offsets, n_reads, total_bases, max_read_len, min_mapq,
                        d_assign, d_best=0, d_nhits=0, d_counts=0):
        """All arguments are raw device pointers (ints); asynchronous on the engine stream."""
        check(lib().mnc_classify_device(self._h, d_bases, d_offsets, n_reads, total_bases, max_read_len,
                                        min_mapq, d_assign, d_best or None, d_nhits or None,
                                        d_counts or None))
        self.n_reads = n_reads

    def sync(self):
        check(lib().mnc_engine_sync(self._h))

    def fetch_hits(self):
        """Gated hit lists of the last batch as CSR (offsets[n+1], hits)."""
        n = C.c_int64(0)
        off = np.zeros(self.n_reads + 1, dtype=np.int64)
        rc = lib().mnc_engine_fetch_hits(self._h, None, None, 0, C.byref(n))
        if rc not in (OK, ERR_RANGE):
            check(rc)
        hits = np.zeros(max(n.value, 1), dtype=HIT_DTYPE)
        check(lib().mnc_engine_fetch_hits(self._h, off.ctypes.data, hits.ctypes.data, len(hits), C.byref(n)))
        return off, hits[:n.value]

    def set_profiling(self, on=True):
        check(lib().mnc_engine_set_profiling(self._h, 1 if on else 0))

    def set_contract(self, contract):
        """CONTRACT_DP (default): base-level alignment as mappy runs it; CONTRACT_CHAIN: stop after chaining."""
        check(lib().mnc_engine_set_contract(self._h, int(contract)))

    def cigars(self):
        """CIGARs [(len, op), ...] of the regions of the last batch, in DUMP_REGS order."""
        regs = self.dump(DUMP_REGS, REG_DTYPE)
        words = self.dump(DUMP_CIGARS, np.uint32)
        out, k = [], 0
        for r in regs:
            n = int(r["n_cigar"])
            out.append([(int(c) >> 4, "MID"[int(c) & 0xf]) for c in words[k:k + n]])
            k += n
        return out

    def set_debug(self, mode=1):
        """mode 2 = stress build of the chaining kernel (every look-back through HBM)."""
        check(lib().mnc_engine_set_debug(self._h, int(mode)))

    def dump_tables(self):
        """The device tables of this engine's index as bytes (test hook)."""
        n = C.c_int64(0)
        rc = lib().mnc_engine_dump_tables(self._h, None, 0, C.byref(n))
        if rc not in (OK, ERR_RANGE):
            check(rc)
        buf = np.zeros(n.value, dtype=np.uint8)
        check(lib().mnc_engine_dump_tables(self._h, buf.ctypes.data, len(buf), C.byref(n)))
        return buf

    def timings(self, reset=False):
        ms = np.zeros(N_STAGES, dtype=np.float64)
        ln = np.zeros(N_STAGES, dtype=np.int64)
        check(lib().mnc_engine_get_timings(self._h, ms.ctypes.data, ln.ctypes.data, 1 if reset else 0))
        names = [lib().mnc_stage_name(s).decode() for s in range(N_STAGES)]
        return {names[s]: (float(ms[s]), int(ln[s])) for s in range(N_STAGES)}

    def counters(self):
        c = np.zeros(24, dtype=np.int64)
        check(lib().mnc_engine_get_counters(self._h, c.ctypes.data, 24))
        keys = ["minimizers", "probe_hits", "anchors", "chains", "regions", "gated_hits", "ambiguous_reads", "batch_redone",
                "dp_segments", "dp_fill_tier1", "dp_fill_tier2", "dp_fill_handed_back",
                "dp_fill_steps_t1", "dp_fill_steps_t2", "dp_fill_steps_t3", "dp_ext_cell_steps",
                "dp_literal_big", "dp_literal_mid", "dp_long_gaps", "dp_literal_big_handed_back", "dp_long_extensions", "dp_fill_tier3",
                "dp_fill_tier_mid", "dp_fill_steps_tm"]
        return dict(zip(keys, (int(x) for x in c)))

    def dump(self, what, dtype):
        n = C.c_int64(0)
        rc = lib().mnc_engine_dump(self._h, what, None, 0, C.byref(n))
        if rc not in (OK, ERR_RANGE):
            check(rc)
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        check(lib().mnc_engine_dump(self._h, what, buf.ctypes.data, len(buf), C.byref(n)))
        return buf[:n.value].view(dtype).copy()


# ---------------------------------------------------------------------------------- host helpers
def best_hit(hits):
    """Exact-integer restatement of aligner.best_hit over (nm, mlen) pairs -> index or -1."""
    arr = np.zeros(len(hits), dtype=HIT_DTYPE)
    for i, (nm, mlen) in enumerate(hits):
        arr[i]["nm"], arr[i]["mlen"] = nm, mlen
    out = C.c_int(0)
    check(lib().mnc_best_hit(arr.ctypes.data, len(hits), C.byref(out)))
    return out.value


def counts(index, assign, best, offsets, mode):
    """Host accumulation of aligner.py:247-263 into an int64[n_genomes] vector."""
    assign = np.ascontiguousarray(assign, dtype=np.int32)
    best = np.ascontiguousarray(best, dtype=HIT_DTYPE)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    out = np.zeros(len(index.genome_names), dtype=np.int64)
    check(lib().mnc_counts(index._h, assign.ctypes.data, best.ctypes.data, offsets.ctypes.data,
                           len(assign), mode, out.ctypes.data))
    return out


# ---------------------------------------------------------------------------------- FASTQ + carried hits
TO_UNMAPPED, TO_AMBIGUOUS, TO_MAPPED, TO_FOCUS = 1, 2, 4, 8


class FastqReader:
    """Batches of a FASTQ file as flat arrays held by the library (``mnc_fastq``).  A malformed
    file raises ValueError with Biopython's message, as SeqIO.parse would."""

    def __init__(self, path, _handle=None, _n=0):
        if _handle is not None:                     # a detached batch (see detach)
            self._h, self.n = _handle, _n
            return
        h = C.c_void_p()
        check(lib().mnc_fastq_open(_b(path), C.byref(h)))
        self._h = h
        self.n = 0

    def remaining(self):
        """Bytes of the file that no batch has taken yet (-1 when unknown)."""
        n = C.c_int64(-1)
        check(lib().mnc_fastq_remaining(self._h, C.byref(n)))
        return n.value

    def detach(self):
        """The current batch as an object of its own; this reader goes on with the next batch in fresh
        arrays.  The detached batch has the reader's accessors, `route`, and is what HitMap.update takes."""
        h = C.c_void_p()
        check(lib().mnc_fastq_detach_batch(self._h, C.byref(h)))
        out = FastqReader(None, _handle=h, _n=self.n)
        self.n = 0
        return out

    def next(self, max_reads=100_000, max_bases=1 << 29):
        n = C.c_uint32(0)
        rc = lib().mnc_fastq_next(self._h, max_reads, max_bases, C.byref(n))
        if rc == ERR_FORMAT:
            raise ValueError(lib().mnc_last_error().decode(errors="replace"))
        check(rc)
        self.n = n.value
        return self.n

    @property
    def bases_ptr(self):
        return lib().mnc_fastq_bases(self._h)

    @property
    def offsets_ptr(self):
        return lib().mnc_fastq_offsets(self._h)

    def offsets(self):
        """int64[n + 1] view, valid until the next batch."""
        return np.ctypeslib.as_array(C.cast(self.offsets_ptr, C.POINTER(C.c_int64)), shape=(self.n + 1,))

    def bases(self):
        total = int(self.offsets()[-1])
        if total == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(self.bases_ptr, C.POINTER(C.c_uint8)), shape=(total,))

    def quals(self):
        total = int(self.offsets()[-1])
        if total == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(lib().mnc_fastq_quals(self._h), C.POINTER(C.c_uint8)), shape=(total,))

    def title(self, r):
        p, ln, idl = C.c_void_p(), C.c_uint32(0), C.c_uint32(0)
        check(lib().mnc_fastq_title(self._h, r, C.byref(p), C.byref(ln), C.byref(idl)))
        return C.string_at(p.value, ln.value).decode(errors="replace") if ln.value else ""

    def route(self, dest, label=None, labels=(), paths=(None, None, None, None)):
        """Append the batch's records to paths = (unmapped, ambiguous, mapped, focus) by dest bits."""
        dest = np.ascontiguousarray(dest, dtype=np.uint8)
        assert len(dest) == self.n
        lab = np.ascontiguousarray(label, dtype=np.int32) if label is not None else None
        bl = [_b(x) for x in labels]
        al = (C.c_char_p * max(len(bl), 1))(*bl)
        ap = (C.c_char_p * 4)(*[(_b(x) if x is not None else None) for x in paths])
        check(lib().mnc_fastq_route(self._h, dest.ctypes.data, lab.ctypes.data if lab is not None else None,
                                    al, len(bl), ap))

    def close(self):
        if getattr(self, "_h", None):
            lib().mnc_fastq_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HitMap:
    """`sample_hits` of the reference as per-id summaries (``mnc_hitmap``)."""

    def __init__(self, path=None):
        h = C.c_void_p()
        if path is None:
            check(lib().mnc_hitmap_create(C.byref(h)))
        else:
            check(lib().mnc_hitmap_load(_b(path), C.byref(h)))
        self._h = h

    def save(self, path):
        check(lib().mnc_hitmap_save(self._h, _b(path)))

    def __len__(self):
        return int(lib().mnc_hitmap_size(self._h))

    def update(self, reader, index, assign, best, nhits):
        """Extend every read's list by this part's hits; returns int32[n, 5] =
        {hits, nm, mlen, name id, tied} of the lists as they stand."""
        out = np.zeros((reader.n, 5), dtype=np.int32)
        assign = np.ascontiguousarray(assign, dtype=np.int32)
        best = np.ascontiguousarray(best, dtype=HIT_DTYPE)
        nhits = np.ascontiguousarray(nhits, dtype=np.int32)
        check(lib().mnc_hitmap_update(self._h, reader._h, index._h, assign.ctypes.data, best.ctypes.data,
                                      nhits.ctypes.data, out.ctypes.data))
        return out

    def names(self):
        L = lib()
        return [L.mnc_hitmap_name(self._h, i).decode() for i in range(L.mnc_hitmap_n_names(self._h))]

    def close(self):
        if getattr(self, "_h", None):
            lib().mnc_hitmap_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
