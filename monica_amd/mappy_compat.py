"""Drop-in for the slice of `mappy` that monica consumes.

The reference uses exactly this surface (monica/genomes/aligner.py):

* `mappy.Aligner(fn_idx_in=<fasta.gz>, preset='map-ont', best_n=15, fn_idx_out=<path>)`  (45-46)
* `mappy.Aligner(fn_idx_in=<index file>)`                                                  (59)
* truthiness of the object (`if not index: raise ...`)                                     (47, 60)
* `index.map(str(seq))` yielding hits with `.is_primary .mapq .ctg .NM .mlen`              (193-195, 215-217)

`Aligner.map` is the per-read slow path kept for fidelity; `Aligner.map_batch` is what
`monica_amd.aligner.aligner` uses: one C-ABI call per micro-batch, all compute on the GPU.
Hits carry what mappy computes with its always-on base-level alignment (MM_F_CIGAR): `mapq` from
the DP branch of minimap2's formula, `mlen` / `blen` / `NM = blen - mlen + n_ambi` from the CIGAR
(the restatement is unpinned against the real library, see there).
"""
import os
import sys
import threading

import numpy as np

from . import _capi


# monica re-invokes `index_loader` on every pass of its real-time loop (aligner.py:56-62 from
# monica.py:437-439); the loaded index and its device-resident tables are kept across those
# calls, keyed by the file's identity, so a pass does not pay the load and upload again.
_INDEX_CACHE = {}
# re-entrant: `Aligner.__del__` takes it, and the collector may run a finaliser on the thread that already holds it
# (any allocation inside a locked region can start a collection); the locked regions below iterate over snapshots
_INDEX_CACHE_LOCK = threading.RLock()
_INDEX_CACHE_MAX = 4
# ... and by bytes: the reference holds ONE part at a time (aligner.py:91-103 rebinds `index` per part); a
# database of many parts kept resident would exhaust host memory or HBM.  Host side: the file sizes of the
# cached parts against a share of physical memory; device side: what the parts' tables and the idle engines'
# batch buffers hold against a share of the device's HBM (the alignment scratch is one per device and is not
# counted).  The most recently used part always stays.
_HOST_SHARE, _DEVICE_SHARE = 0.5, 0.6
_CACHE_BYTES = {}                    # key -> file size (the in-memory form of a part is about its file)
# idle engines (stream + per-batch HBM buffers), per DEVICE: an engine is not tied to an index part -- it is
# rebound to the part at hand (`mnc_engine_set_index`), as the reference's threads keep running while `index` is
# rebound in the loop over the parts (aligner.py:91-103).  So a database of many parts needs as many engines as
# monica has threads, not threads x parts.  (Kept outside the Index objects: an Engine refers to its Index.)
_ENGINE_POOLS = {}
_POOL_MAX_PER_DEVICE = 8


def reserve_index_cache(n_parts):
    """Make room for every part of a multi-part database (monica loops over all parts on every
    pass, aligner.py:91-103: a cache smaller than the loop would miss on every load).  The byte
    bounds still hold: parts that do not fit together are loaded again, as the reference does."""
    global _INDEX_CACHE_MAX
    with _INDEX_CACHE_LOCK:
        _INDEX_CACHE_MAX = max(_INDEX_CACHE_MAX, int(n_parts))


def _host_budget():
    try:
        return int(os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") * _HOST_SHARE)
    except (ValueError, OSError):
        return 1 << 62


def _device_budget(device):
    try:
        return int(_capi.device_mem_info(device)[1] * _DEVICE_SHARE)
    except _capi.MncError:
        return 1 << 62                                         # no device: nothing is resident either


def _device_bytes_locked(device):
    # (a closed handle holds nothing)
    n = sum(index.device_bytes(device) for index in list(_INDEX_CACHE.values()) if getattr(index, "_h", True))
    return n + sum(eng.device_bytes() for eng in list(_ENGINE_POOLS.get(device, [])) if getattr(eng, "_h", True))


def _drop_locked(key):
    old = _INDEX_CACHE.pop(key)
    _CACHE_BYTES.pop(key, None)
    for pool in _ENGINE_POOLS.values():                        # idle engines still bound to the part would keep it alive
        for eng in [e for e in pool if e.index is old]:
            pool.remove(eng)
            eng.close()
    # the Index frees its host and device tables when the last Aligner using it lets go


def _evict_locked(device=None):
    """Least recently used parts go until the count and both byte budgets hold (the newest stays)."""
    while len(_INDEX_CACHE) > _INDEX_CACHE_MAX:
        _drop_locked(next(iter(_INDEX_CACHE)))
    host = _host_budget()
    while len(_INDEX_CACHE) > 1 and sum(_CACHE_BYTES.values()) > host:
        _drop_locked(next(iter(_INDEX_CACHE)))
    if device is not None:
        dev = _device_budget(device)
        while len(_INDEX_CACHE) > 1 and _device_bytes_locked(device) > dev:
            _drop_locked(next(iter(_INDEX_CACHE)))
        pool = _ENGINE_POOLS.get(device, [])
        while pool and _device_bytes_locked(device) > dev:      # one part alone and still over: the idle engines go
            pool.pop().close()


def release_idle(keep_index=None):
    """Free what the cache holds beyond `keep_index`: every idle engine and every other part.
    Called when an allocation fails (MNC_ERR_NOMEM) before the one retry."""
    with _INDEX_CACHE_LOCK:
        for key in [k for k, v in _INDEX_CACHE.items() if v is not keep_index]:
            _drop_locked(key)
        for pool in _ENGINE_POOLS.values():
            while pool:
                pool.pop().close()


def _load_index_cached(path, device=None):
    st = os.stat(path)
    key = (os.path.realpath(path), st.st_mtime_ns, st.st_size)
    with _INDEX_CACHE_LOCK:
        hit = _INDEX_CACHE.get(key)
        if hit is not None:
            _INDEX_CACHE[key] = _INDEX_CACHE.pop(key)          # most recently used last
            return hit
    try:
        index = _capi.Index.load(path)
    except _capi.MncError as e:
        if e.code != _capi.ERR_NOMEM:
            raise
        release_idle()
        index = _capi.Index.load(path)
    with _INDEX_CACHE_LOCK:
        _INDEX_CACHE[key] = index
        _CACHE_BYTES[key] = st.st_size
        _evict_locked(device)
    return index


def _is_cached(index):
    return any(v is index for v in _INDEX_CACHE.values())


def default_device():
    for var in ("MONICA_AMD_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v is not None and v.strip().lstrip("-").isdigit():
            return int(v)
    return 0


def _build_device(device):
    """The device that sketches and sorts an index being built (csrc/k_idxbuild.hip), or None for the host builder
    when no device is visible (building an index is not part of the classified path; both give the same index)."""
    try:
        return device if 0 <= device < _capi.device_count() else None
    except _capi.MncError:
        return None


class Hit:
    """The attributes of a mappy alignment."""
    __slots__ = ("ctg", "ctg_len", "r_st", "r_en", "q_st", "q_en", "strand", "mapq", "mlen", "blen", "NM",
                 "is_primary", "score", "n_anchors", "cigar")

    def __init__(self, reg, index, cigar=None):
        rid = int(reg["rid"])
        self.ctg = index.contig_names[rid]
        self.ctg_len = index.contig_lens[rid]
        self.r_st, self.r_en = int(reg["rs"]), int(reg["re"])
        self.q_st, self.q_en = int(reg["qs"]), int(reg["qe"])
        self.strand = -1 if reg["rev"] else 1
        self.mapq = int(reg["mapq"])
        self.mlen, self.blen = int(reg["mlen"]), int(reg["blen"])
        self.NM = self.blen - self.mlen + int(reg["n_ambi"])
        self.cigar = [[l, "MID".index(op)] for l, op in cigar] if cigar else []      # mappy: [[length, op], ...]
        self.is_primary = bool(reg["id"] == reg["parent"])
        self.score = int(reg["score"])
        self.n_anchors = int(reg["cnt"])

    @property
    def cigar_str(self):
        return "".join(f"{l}{'MID'[op]}" for l, op in self.cigar)

    def __repr__(self):
        return (f"{self.q_st}\t{self.q_en}\t{'+' if self.strand > 0 else '-'}\t{self.ctg}\t{self.ctg_len}\t"
                f"{self.r_st}\t{self.r_en}\t{self.mlen}\t{self.blen}\t{self.mapq}\t"
                f"tp:A:{'P' if self.is_primary else 'S'}")


# What `fn_idx_out` writes: "native" (this library's file: loads without sorting) or "mmi" (minimap2's format, as
# mappy writes it at aligner.py:45-46 -- for an installation that also runs the reference).  Both load.
INDEX_FILE_FORMAT = "native"


class Aligner:
    """Index handle with mappy's constructor and truthiness."""

    def __init__(self, fn_idx_in=None, preset=None, k=None, w=None, min_cnt=None, min_chain_score=None,
                 min_dp_score=None, bw=None, best_n=None, n_threads=3, fn_idx_out=None, max_frag_len=None,
                 extra_flags=None, seq=None, scoring=None, device=None):
        self._index = None
        self._borrowed = []
        self._tls = threading.local()
        self._device = default_device() if device is None else int(device)
        self.error = None
        kk = 15 if k is None else int(k)
        ww = 10 if w is None else int(w)
        try:
            if seq is not None:
                self._index = _capi.Index.from_seqs(["N/A"], [seq], kk, ww)
            elif fn_idx_in is None:
                raise ValueError("fn_idx_in or seq is required")
            elif self._is_index_file(fn_idx_in):
                self._index = _load_index_cached(fn_idx_in, self._device)
            elif INDEX_FILE_FORMAT == "mmi" and fn_idx_out:
                self._index = _capi.Index.build(fn_idx_in, None, kk, ww, device=_build_device(self._device))
                self._index.save(fn_idx_out, mmi=True)
            else:
                self._index = _capi.Index.build(fn_idx_in, fn_idx_out, kk, ww, device=_build_device(self._device))
        except (_capi.MncError, OSError, ValueError) as e:      # mappy: a falsy Aligner, no exception ...
            self.error = e
            self._index = None
            print(f"[monica_amd] {fn_idx_in}: {type(e).__name__}: {e}", file=sys.stderr)     # ... and the C layer's line on stderr

    @staticmethod
    def _is_index_file(path):
        try:
            with open(path, "rb") as f:
                head = f.read(6)
                return head == b"MNCIDX" or head[:4] == b"MMI\x02"       # this library's file, or minimap2's own
        except OSError:
            return False

    def __bool__(self):
        return self._index is not None

    @property
    def index(self):
        return self._index

    @property
    def k(self):
        return self._index.k

    @property
    def w(self):
        return self._index.w

    @property
    def n_seq(self):
        return len(self._index.contig_names)

    @property
    def seq_names(self):
        return list(self._index.contig_names)

    def engine(self):
        """One engine (HIP stream + HBM batch buffers) per calling thread: the reference shares one
        index between the threads of its pool (aligner.py:89-103).  Engines outlive this object and
        are not tied to an index part: an idle engine of the device is rebound to this part."""
        e = getattr(self._tls, "engine", None)
        if e is None:
            with _INDEX_CACHE_LOCK:
                pool = _ENGINE_POOLS.setdefault(self._device, [])
                e = pool.pop() if pool else None
            if e is not None:
                try:
                    e.set_index(self._index)
                except _capi.MncError:                         # made for another k / w: not reusable here
                    e.close()
                    e = None
            if e is None:
                try:
                    e = _capi.Engine(self._index, self._device)
                except _capi.MncError as err:                  # HBM exhausted: free what is idle, once
                    if err.code != _capi.ERR_NOMEM:
                        raise
                    release_idle(self._index)
                    e = _capi.Engine(self._index, self._device)
            self._tls.engine = e
            with _INDEX_CACHE_LOCK:
                self._borrowed.append(e)
        return e

    def __del__(self):
        try:
            if self._index is not None and self._borrowed:
                with _INDEX_CACHE_LOCK:
                    pool = _ENGINE_POOLS.setdefault(self._device, [])
                    # only engines of a cached part go back: one bound to an uncached index would keep that index alive
                    room = max(0, _POOL_MAX_PER_DEVICE - len(pool)) if _is_cached(self._index) else 0
                    keep, rest = self._borrowed[:room], self._borrowed[room:]
                    pool.extend(keep)
                    self._borrowed = []
                for eng in rest:                               # not pooled: release the HBM now
                    eng.close()
                with _INDEX_CACHE_LOCK:                        # the returned engines count against the device budget
                    _evict_locked(self._device)
        except Exception:
            pass

    # ------------------------------------------------------------------ batched fast path
    def map_batch(self, bases, offsets, min_mapq=0):
        """Classify one micro-batch.  Returns (assign, best, nhits, hit_offsets, hits) where
        hits[hit_offsets[r]:hit_offsets[r+1]] are the (rid, mapq, nm, mlen) of the hits of read r
        that pass `is_primary and mapq >= min_mapq`."""
        eng = self.engine()
        try:
            assign, best, nhits = eng.classify(bases, offsets, min_mapq)
        except _capi.MncError as err:                          # the batch's buffers did not fit beside the cache
            if err.code != _capi.ERR_NOMEM:
                raise
            release_idle(self._index)
            assign, best, nhits = eng.classify(bases, offsets, min_mapq)
        hit_off, hits = eng.fetch_hits()
        return assign, best, nhits, hit_off, hits

    # ------------------------------------------------------------------ per-read compatibility path
    def map(self, seq, seq2=None, buf=None, cs=False, MD=False, max_frag_len=None, extra_flags=None):
        if self._index is None:
            return
        b = seq if isinstance(seq, (bytes, bytearray)) else str(seq).encode()
        eng = self.engine()
        eng.classify(np.frombuffer(b, dtype=np.uint8), np.array([0, len(b)], dtype=np.int64), 0)
        regs = eng.dump(_capi.DUMP_REGS, _capi.REG_DTYPE)
        cigs = eng.cigars()
        for reg, cig in zip(regs, cigs):
            yield Hit(reg, self._index, cig)
