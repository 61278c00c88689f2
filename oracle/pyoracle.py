"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Loads ``oracle/liborc.so`` (built by ``oracle/Makefile`` from ``mm_oracle.c``).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``monica_amd`` never does.

PARITY UNPINNED: see the header of ``mm_oracle.h``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

UNMAPPED = -1
AMBIGUOUS = -2


class Opt(C.Structure):
    _fields_ = [("seed", C.c_int), ("mid_occ_frac", C.c_float), ("mid_occ", C.c_int),
                ("min_cnt", C.c_int), ("min_chain_score", C.c_int), ("bw", C.c_int),
                ("max_gap", C.c_int), ("max_chain_skip", C.c_int), ("max_chain_iter", C.c_int),
                ("mask_level", C.c_float), ("pri_ratio", C.c_float), ("best_n", C.c_int),
                ("max_join_long", C.c_int), ("max_join_short", C.c_int),
                ("min_join_flank_sc", C.c_int), ("min_join_flank_ratio", C.c_float),
                ("a", C.c_int), ("b", C.c_int),
                ("cigar", C.c_int), ("q", C.c_int), ("e", C.c_int), ("q2", C.c_int), ("e2", C.c_int),
                ("sc_ambi", C.c_int), ("zdrop", C.c_int), ("zdrop_inv", C.c_int), ("end_bonus", C.c_int),
                ("min_dp_max", C.c_int), ("min_ksw_len", C.c_int), ("max_clip_ratio", C.c_float),
                ("max_sw_mat", C.c_int64)]


class Extz(C.Structure):
    _fields_ = [("max", C.c_uint32), ("zdropped", C.c_int), ("max_q", C.c_int), ("max_t", C.c_int),
                ("mqe", C.c_int), ("mqe_t", C.c_int), ("mte", C.c_int), ("mte_q", C.c_int),
                ("score", C.c_int), ("m_cigar", C.c_int), ("n_cigar", C.c_int), ("reach_end", C.c_int),
                ("cigar", C.POINTER(C.c_uint32))]


EZ_RIGHT, EZ_APPROX_MAX, EZ_EXTZ_ONLY, EZ_REV_CIGAR = 0x02, 0x08, 0x40, 0x80


REG_DTYPE = np.dtype([("id", "<i4"), ("parent", "<i4"), ("rid", "<i4"), ("rev", "<i4"),
                      ("rs", "<i4"), ("re", "<i4"), ("qs", "<i4"), ("qe", "<i4"),
                      ("score", "<i4"), ("score0", "<i4"), ("cnt", "<i4"), ("as", "<i4"),
                      ("mlen", "<i4"), ("blen", "<i4"), ("subsc", "<i4"), ("n_sub", "<i4"),
                      ("mapq", "<i4"), ("hash", "<u4"),
                      ("dp_score", "<i4"), ("dp_max", "<i4"), ("dp_max2", "<i4"), ("n_ambi", "<i4"),
                      ("n_cigar", "<i4"), ("flags", "<i4")])
HIT_DTYPE = np.dtype([("rid", "<i4"), ("mapq", "<i4"), ("nm", "<i4"), ("mlen", "<i4")])
A128_DTYPE = np.dtype([("x", "<u8"), ("y", "<u8")])


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = [os.path.join(_HERE, f) for f in ("mm_oracle.c", "mm_ksw.c", "mm_align.c", "mm_oracle.h", "mm_internal.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, i32, i64, u64, cp = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_char_p
        L.orc_opt_init.argtypes = [C.POINTER(Opt)]
        L.orc_hash64.restype = u64
        L.orc_hash64.argtypes = [u64, u64]
        L.orc_sketch.restype = i32
        L.orc_sketch.argtypes = [cp, i32, i32, i32, C.c_uint32, vp, i32]
        L.orc_index_build_mem.restype = vp
        L.orc_index_build_mem.argtypes = [i32, C.POINTER(cp), C.POINTER(cp), C.POINTER(i32), i32, i32]
        L.orc_index_build_fasta.restype = vp
        L.orc_index_build_fasta.argtypes = [cp, i32, i32]
        L.orc_index_free.argtypes = [vp]
        for f in ("orc_index_k", "orc_index_w", "orc_index_n_seq"):
            getattr(L, f).restype = i32
            getattr(L, f).argtypes = [vp]
        L.orc_index_name.restype = cp
        L.orc_index_name.argtypes = [vp, i32]
        L.orc_index_len.restype = i32
        L.orc_index_len.argtypes = [vp, i32]
        L.orc_index_n_minimizers.restype = i64
        L.orc_index_n_minimizers.argtypes = [vp]
        L.orc_index_n_keys.restype = i64
        L.orc_index_n_keys.argtypes = [vp]
        L.orc_index_cal_mid_occ.restype = i32
        L.orc_index_cal_mid_occ.argtypes = [vp, C.c_float]
        L.orc_index_dump.restype = i64
        L.orc_index_dump.argtypes = [vp, vp, vp, i64]
        L.orc_collect_seeds.restype = i64
        L.orc_collect_seeds.argtypes = [vp, C.POINTER(Opt), i32, cp, i32, C.POINTER(vp), C.POINTER(i32)]
        L.orc_chain_dp.restype = vp
        L.orc_chain_dp.argtypes = [C.POINTER(Opt), i64, vp, C.POINTER(i32), C.POINTER(vp), vp, vp, vp]
        L.orc_map.restype = i32
        L.orc_map.argtypes = [vp, C.POINTER(Opt), i32, cp, i32, vp, i32]
        L.orc_best_hit.restype = i32
        L.orc_best_hit.argtypes = [vp, i32]
        L.orc_classify_batch.restype = i64
        L.orc_classify_batch.argtypes = [vp, C.POINTER(Opt), i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, i64]
        i8p = C.c_void_p
        L.orc_gen_simple_mat.argtypes = [i32, i8p, C.c_int8, C.c_int8, C.c_int8]
        L.orc_ksw_extd2.argtypes = [i32, vp, i32, vp, C.c_int8, i8p, C.c_int8, C.c_int8, C.c_int8, C.c_int8,
                                    i32, i32, i32, i32, C.POINTER(Extz)]
        L.orc_dp_clean.argtypes = [i32, vp, i32, vp, i8p, i32, i32, i32, i32, i32, i32, i32, C.POINTER(Extz)]
        L.orc_local_score.restype = i32
        L.orc_local_score.argtypes = [i32, vp, i32, vp, i8p, i32, i32]
        L.orc_map_cigar.restype = i32
        L.orc_map_cigar.argtypes = [vp, C.POINTER(Opt), i32, cp, i32, vp, i32, vp, i32, C.POINTER(i32)]
        L._libc = C.CDLL(None)
        L._libc.free.argtypes = [vp]
        _LIB = L
    return _LIB


def simple_mat(a=2, b=4, sc_ambi=1):
    mat = np.zeros(25, dtype=np.int8)
    lib().orc_gen_simple_mat(5, mat.ctypes.data, a, b, sc_ambi)
    return mat


def _ez_result(ez):
    cig = [(int(ez.cigar[i]) >> 4, "MID"[int(ez.cigar[i]) & 0xf]) for i in range(ez.n_cigar)]
    out = dict(max=int(ez.max), zdropped=int(ez.zdropped), max_q=ez.max_q, max_t=ez.max_t, mqe=ez.mqe, mqe_t=ez.mqe_t,
               mte=ez.mte, mte_q=ez.mte_q, score=ez.score, reach_end=ez.reach_end, cigar=cig)
    if ez.cigar:
        lib()._libc.free(C.cast(ez.cigar, C.c_void_p))
    return out


def ksw_extd2(query, target, w=751, zdrop=400, end_bonus=-1, flag=0, q=4, e=2, q2=24, e2=1, mat=None):
    """The literal simulation of ksw2's kernel on code arrays (0..4)."""
    mat = simple_mat() if mat is None else mat
    qa = np.ascontiguousarray(query, dtype=np.uint8)
    ta = np.ascontiguousarray(target, dtype=np.uint8)
    ez = Extz()
    lib().orc_ksw_extd2(len(qa), qa.ctypes.data, len(ta), ta.ctypes.data, 5, mat.ctypes.data, q, e, q2, e2,
                        w, zdrop, end_bonus, flag, C.byref(ez))
    return _ez_result(ez)


def dp_clean(query, target, zdrop=400, end_bonus=-1, flag=0, q=4, e=2, q2=24, e2=1, mat=None):
    """The same recurrence in absolute scores, unbanded."""
    mat = simple_mat() if mat is None else mat
    qa = np.ascontiguousarray(query, dtype=np.uint8)
    ta = np.ascontiguousarray(target, dtype=np.uint8)
    ez = Extz()
    lib().orc_dp_clean(len(qa), qa.ctypes.data, len(ta), ta.ctypes.data, mat.ctypes.data, q, e, q2, e2,
                       zdrop, end_bonus, flag, C.byref(ez))
    return _ez_result(ez)


def default_opt():
    o = Opt()
    lib().orc_opt_init(C.byref(o))
    return o


def hash64(key, mask):
    return int(lib().orc_hash64(int(key), int(mask)))


def _b(seq):
    return seq if isinstance(seq, (bytes, bytearray)) else seq.encode()


def sketch(seq, w=10, k=15, rid=0):
    """Minimizers of one sequence as a structured array (x = hash<<8|span, y = rid<<32|pos<<1|strand)."""
    s = _b(seq)
    cap = max(len(s), 1)
    out = np.zeros(cap, dtype=A128_DTYPE)
    n = lib().orc_sketch(s, len(s), w, k, rid, out.ctypes.data, cap)
    return out[:n].copy()


class Index:
    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle index build failed")
        self._h = handle
        L = lib()
        self.k = L.orc_index_k(handle)
        self.w = L.orc_index_w(handle)
        self.n_seq = L.orc_index_n_seq(handle)
        self.names = [L.orc_index_name(handle, i).decode() for i in range(self.n_seq)]
        self.lens = [L.orc_index_len(handle, i) for i in range(self.n_seq)]
        self.opt = default_opt()
        self.mid_occ = L.orc_index_cal_mid_occ(handle, self.opt.mid_occ_frac)

    @classmethod
    def from_seqs(cls, names, seqs, k=15, w=10):
        n = len(names)
        bn = [_b(x) for x in names]
        bs = [_b(x) for x in seqs]
        an = (C.c_char_p * n)(*bn)
        as_ = (C.c_char_p * n)(*bs)
        al = (C.c_int * n)(*[len(x) for x in bs])
        return cls(lib().orc_index_build_mem(n, an, as_, al, k, w))

    @classmethod
    def from_fasta(cls, path, k=15, w=10):
        return cls(lib().orc_index_build_fasta(_b(path), k, w))

    def __del__(self):
        try:
            if self._h:
                lib().orc_index_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def n_minimizers(self):
        return int(lib().orc_index_n_minimizers(self._h))

    @property
    def n_keys(self):
        return int(lib().orc_index_n_keys(self._h))

    def dump(self):
        n = self.n_minimizers
        h = np.zeros(n, dtype=np.uint64)
        y = np.zeros(n, dtype=np.uint64)
        lib().orc_index_dump(self._h, h.ctypes.data, y.ctypes.data, n)
        return h, y

    def seeds(self, seq):
        """Sorted anchors of one read + rep_len (A.4)."""
        s = _b(seq)
        pa = C.c_void_p()
        rep = C.c_int()
        n = lib().orc_collect_seeds(self._h, C.byref(self.opt), self.mid_occ, s, len(s),
                                    C.byref(pa), C.byref(rep))
        if n > 0:
            buf = (C.c_char * (16 * n)).from_address(pa.value)
            out = np.frombuffer(buf, dtype=A128_DTYPE).copy()
        else:
            out = np.zeros(0, dtype=A128_DTYPE)
        if pa.value:
            lib()._libc.free(pa)
        return out, rep.value

    def chain(self, seq):
        """DP arrays (f, p, v) over the sorted anchors, plus chains u[] and the reordered anchors."""
        a, rep = self.seeds(seq)
        n = len(a)
        f = np.zeros(n, dtype=np.int32)
        p = np.zeros(n, dtype=np.int32)
        v = np.zeros(n, dtype=np.int32)
        if n == 0:
            return a, f, p, v, np.zeros(0, dtype=np.uint64), a
        L = lib()
        raw = L._libc
        raw.malloc.restype = C.c_void_p
        raw.malloc.argtypes = [C.c_size_t]
        pa = raw.malloc(16 * n)
        C.memmove(pa, a.ctypes.data, 16 * n)
        n_u = C.c_int()
        pu = C.c_void_p()
        pb = L.orc_chain_dp(C.byref(self.opt), n, pa, C.byref(n_u), C.byref(pu),
                            f.ctypes.data, p.ctypes.data, v.ctypes.data)
        if pb and n_u.value > 0:
            u = np.frombuffer((C.c_char * (8 * n_u.value)).from_address(pu.value), dtype=np.uint64).copy()
            tot = int((u & np.uint64(0xffffffff)).sum())
            b = np.frombuffer((C.c_char * (16 * tot)).from_address(pb), dtype=A128_DTYPE).copy()
        else:
            u = np.zeros(0, dtype=np.uint64)
            b = np.zeros(0, dtype=A128_DTYPE)
        if pb:
            raw.free(pb)
        if pu.value:
            raw.free(pu)
        return a, f, p, v, u, b

    def map(self, seq):
        """All kept regions of one read (primary and secondary) as a structured array."""
        s = _b(seq)
        cap = 64
        while True:
            out = np.zeros(cap, dtype=REG_DTYPE)
            n = lib().orc_map(self._h, C.byref(self.opt), self.mid_occ, s, len(s), out.ctypes.data, cap)
            if n <= cap:
                return out[:n].copy()
            cap = n

    def map_cigar(self, seq):
        """Regions of one read and their CIGARs [(len, op), ...] (base-level alignment on)."""
        s = _b(seq)
        cap, ccap = 64, 1 << 16
        while True:
            out = np.zeros(cap, dtype=REG_DTYPE)
            cig = np.zeros(ccap, dtype=np.uint32)
            tot = C.c_int(0)
            n = lib().orc_map_cigar(self._h, C.byref(self.opt), self.mid_occ, s, len(s), out.ctypes.data, cap,
                                    cig.ctypes.data, ccap, C.byref(tot))
            if n <= cap and tot.value <= ccap:
                break
            cap, ccap = max(cap, n), max(ccap, tot.value)
        regs, cigs, k = out[:n].copy(), [], 0
        for r in regs:
            cigs.append([(int(c) >> 4, "MID"[int(c) & 0xf]) for c in cig[k:k + r["n_cigar"]]])
            k += r["n_cigar"]
        return regs, cigs

    def classify(self, bases, offsets, min_mapq=60, n_threads=1):
        """bases: bytes/uint8 array of concatenated reads; offsets: int64[n+1]."""
        bases = np.ascontiguousarray(np.frombuffer(_b(bases), dtype=np.uint8)
                                     if isinstance(bases, (bytes, bytearray, str)) else bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        assign = np.zeros(n, dtype=np.int32)
        best = np.zeros(n, dtype=HIT_DTYPE)
        nh = np.zeros(n, dtype=np.int32)
        cap = max(4 * n, 16)
        while True:
            flat = np.zeros(cap, dtype=HIT_DTYPE)
            tot = lib().orc_classify_batch(self._h, C.byref(self.opt), self.mid_occ,
                                           bases.ctypes.data, offsets.ctypes.data, n, min_mapq, n_threads,
                                           assign.ctypes.data, best.ctypes.data, nh.ctypes.data,
                                           flat.ctypes.data, cap)
            if tot <= cap:
                break
            cap = int(tot)
        return assign, best, nh, flat[:tot].copy()


def best_hit(hits):
    """hits: sequence of (nm, mlen); returns index of the best or -1 (reference returns 0)."""
    arr = np.zeros(len(hits), dtype=HIT_DTYPE)
    for i, (nm, mlen) in enumerate(hits):
        arr[i]["nm"] = nm
        arr[i]["mlen"] = mlen
    return int(lib().orc_best_hit(arr.ctypes.data, len(hits)))
