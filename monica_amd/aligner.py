"""Host-side mirror of `monica.genomes.aligner` (reference: monica/genomes/aligner.py).

Same public names, argument order, defaults, return shapes and on-disk side effects, so
`monica.monica` (monica.py:264, 437-439, 464-467) and `test/test_aligner.py` can call this
module instead of the reference one.  What changes is underneath: `index.map(read)` per read
(aligner.py:193, 215) becomes one C-ABI call per micro-batch into the gfx950 kernels
(`mappy_compat.Aligner.map_batch`), and Biopython record objects become flat arrays.

One deliberate deviation: index files under the reference's `indexN.mmi` names are this library's
own format unless `mappy_compat.INDEX_FILE_FORMAT = "mmi"`; `index_loader` reads both, and minimap2's
own files.  (`n_threads=None` is `ThreadPool(None)`, one worker per core, as in the reference
(aligner.py:89); a worker makes its engine -- a HIP stream and the batch buffers in HBM -- only when it
is handed a sample file, so idle workers cost nothing on the device.)
"""
import itertools
import os
import pickle
import queue
import threading
import time
from collections import Counter
from multiprocessing.dummy import Pool as ThreadPool

import numpy as np

from . import _capi
from . import mappy_compat as mappy


def _monica_root():
    """MONICA_ROOT as the reference resolves it (aligner.py:15-16), or None when monica has
    not been initialised on this machine (the reference would fail at import)."""
    env = os.environ.get("MONICA_ROOT")
    if env:
        return env
    try:
        with open(os.path.join(os.path.expanduser("~"), ".monica", ".root"), "r") as root:
            return root.readline()
    except OSError:
        return None


MONICA_ROOT = _monica_root()
GENOMES_PATH = os.path.join(MONICA_ROOT, "genomes") if MONICA_ROOT else None

BEST_N = 15
INDEXES_PATH = os.path.join(MONICA_ROOT, "indexes") if MONICA_ROOT else None
INDEX_NAME = ["index", ".mmi"]

ALIGNMENT_PICKLE_FILENAME = "alignment.pkl"

MAPPED_FILES_FOLDER = "mapped"
UNMAPPED_FILES_FOLDER = "unmapped"
AMBIGUOUS_FILES_FOLDER = "ambiguous"
HITS_FILES_FOLDER = "hits"
FOCUS_FILES_FOLDER = "focus"

# reads / bases per C-ABI call: a sample file is worked through in batches of this size, three at a time (one being
# parsed, one on the GPU, one being written out)
# (12 500 since round 5: the second call on a 1 GB file over four sweeps -- profiles/r04y2_, r04zz_, r05z_, r05v_files_sweep.txt --
# ran at 0.68 / 0.74 / 0.82 / 0.84 M reads/s with it, at 0.82 / 0.60 / 0.59 / 0.61 M with 25 000 and at 0.58-0.64 M with true
# 50 000-read batches: the stages get cheaper per gigabyte as batches grow, the pipeline's fill and drain cost more than that)
BATCH_READS = int(os.environ.get("MONICA_AMD_BATCH_READS", "12500"))
BATCH_BASES = min(1 << 27, BATCH_READS * 6000)
# the first batches of a file, as fractions of BATCH_READS: the stages behind the parser wait for the first batch, so it is small
BATCH_RAMP = tuple(float(x) for x in os.environ.get("MONICA_AMD_BATCH_RAMP", "0.25").split(",") if x.strip())
DEFAULT_MAX_WORKERS = int(os.environ.get("MONICA_AMD_MAX_WORKERS", "8"))   # `n_threads=None`: at most this many samples in flight
TIMINGS = {}                      # per sample name: seconds spent per phase of aligner() (diagnostics)
PIPE_TRACE = [] if os.environ.get("MONICA_AMD_PIPE_TRACE") else None   # (stage, reads, start, end) per batch and stage (diagnostics)


def _trace(stage, n, t0):
    if PIPE_TRACE is not None:
        PIPE_TRACE.append((stage, int(n), t0, time.perf_counter()))


def _marker(name):
    # progress breadcrumbs of the reference (aligner.py:40, 51)
    if GENOMES_PATH and os.path.isdir(GENOMES_PATH):
        with open(os.path.join(GENOMES_PATH, name), "wb"):
            pass


def indexer(databases, indexes_path=INDEXES_PATH):
    """Build one index per `databaseN.fna.gz` chunk (aligner.py:31-53)."""
    if not os.path.exists(indexes_path):
        os.makedirs(indexes_path)
    else:
        for stale in os.listdir(indexes_path):
            if stale.endswith(".mmi"):
                os.remove(os.path.join(indexes_path, stale))
    built = []
    print("Started building {} index".format(indexes_path))
    _marker("entered_indexer")
    for database in os.listdir(databases):
        if not database.endswith(".fna.gz"):
            continue
        number = os.path.split(database)[1][8:-7]
        target = os.path.join(indexes_path, str(number).join(INDEX_NAME))
        index = mappy.Aligner(fn_idx_in=os.path.join(databases, database), preset="map-ont", best_n=BEST_N,
                              fn_idx_out=target)
        if not index:
            raise Exception("Index building failed") from getattr(index, "error", None)
        built.append(target)
    print("Finished building {} index".format(indexes_path))
    _marker("finished_indexing")
    return built


def index_loader(index_file):
    """Load one index part (aligner.py:56-62)."""
    if index_file.endswith(".mmi"):
        print(f"aligning on {index_file}")
        t0 = time.perf_counter()
        index = mappy.Aligner(fn_idx_in=index_file)
        if not index:
            raise Exception("Damaged or empty index") from getattr(index, "error", None)
        TIMINGS.setdefault("_index_loader", {"load": 0.0})["load"] += time.perf_counter() - t0
        return index


def multi_threaded_aligner(query_folder, indexes_paths, mode=None, mapping_quality=60, overnight=False, n_threads=None,
                           focus_species=[], output_folder=None, mapped_files_folder=MAPPED_FILES_FOLDER,
                           unmapped_files_folder=UNMAPPED_FILES_FOLDER, ambiguous_files_folder=AMBIGUOUS_FILES_FOLDER,
                           hits_files_folder=HITS_FILES_FOLDER, focus_file_folder=FOCUS_FILES_FOLDER):
    """Classify every non-empty `*fastq` file of `query_folder` against the index parts
    (aligner.py:65-111): every part but the last only collects hits; the last part decides."""
    os.chdir(query_folder)
    samples = [f for f in os.listdir(".") if f.endswith("fastq") and os.stat(f).st_size]
    if not samples:
        print("No query files were provided")
        return 0
    samples_name = [s.split(".")[0] for s in samples]

    folders = {k: os.path.join(query_folder, v) for k, v in (
        ("mapped", mapped_files_folder), ("unmapped", unmapped_files_folder), ("ambiguous", ambiguous_files_folder),
        ("hits", hits_files_folder), ("focus", focus_file_folder))}
    if not os.path.exists(folders["mapped"]):
        for key in ("mapped", "unmapped", "ambiguous", "hits"):
            os.mkdir(folders[key])
        if focus_species:
            os.mkdir(folders["focus"])

    # aligner.py:89 is `ThreadPool(n_threads)`; None there means one worker per core, sized for CPU mappers.  Here a worker
    # feeds a GPU: it owns an engine (HBM batch buffers), three pipeline threads and two teams of helper threads (csrc/team.h), and all engines
    # of a device take turns on one alignment workspace -- so an explicit n_threads is kept as given, None is capped at
    # DEFAULT_MAX_WORKERS (and at the number of samples), and the readers' teams share the cores over the workers.
    workers = n_threads if n_threads else min(os.cpu_count() or 1, DEFAULT_MAX_WORKERS)
    workers = max(1, min(int(workers), len(samples)))
    _capi.set_io_workers(workers)
    pool = ThreadPool(workers)
    rep = itertools.repeat
    mappy.reserve_index_cache(len(indexes_paths))
    try:
        for part in indexes_paths[:-1]:
            index = index_loader(part)
            pool.starmap(aligner, zip(samples, samples_name, rep(index), rep(mode), rep(folders["hits"]),
                                      rep(mapping_quality)))
        index = index_loader(indexes_paths[-1])
        _trace("loaded", 0, time.perf_counter())
        results = pool.starmap(aligner, zip(samples, samples_name, rep(index), rep(mode), rep(folders["hits"]),
                                            rep(mapping_quality), rep(overnight), rep(focus_species),
                                            rep(folders["mapped"]), rep(folders["unmapped"]),
                                            rep(folders["ambiguous"]), rep(folders["focus"]), rep(True)))
    finally:
        _trace("mapped", 0, time.perf_counter())
        pool.close()
        _capi.set_io_workers(1)
    t0 = time.perf_counter()
    try:
        return alignment_update(results, output_folder)
    finally:
        _trace("update", 0, t0)


def aligner(sample, sample_name, index, mode=None, hits_folder=None, mapping_quality=None, overnight=False,
            focus_species=[], mapped_folder=None, unmapped_folder=None, ambiguous_folder=None, focus_folder=None,
            last_index=False):
    """One sample file against one index part (aligner.py:179-279).

    Per batch: the library parses the FASTQ records (`mnc_fastq_next`), the GPU classifies them
    (`mnc_classify_batch`), the per-id hit lists carried over from earlier index parts are
    extended (`mnc_hitmap_update`, the reference's `sample_hits`), and on the last part the
    records are appended to mapped/ unmapped/ ambiguous/ focus/ (`mnc_fastq_route`).  Python
    only sees batch-level arrays and the handful of contig names that were hit."""
    # mode parameter is for testing only (reference comment)
    print(f"{sample}, mode is {mode}\t")
    if mapping_quality is None:
        raise TypeError("'>=' not supported between instances of 'int' and 'NoneType'")
    carried_file = os.path.join(hits_folder, sample_name + "_hits.pkl")
    sample_hits = _capi.HitMap(carried_file if os.path.exists(carried_file) else None)
    t_engine = time.perf_counter()
    engine = index.engine()
    t_engine = time.perf_counter() - t_engine
    t_open = time.perf_counter()
    reader = _capi.FastqReader(sample)
    t_open = time.perf_counter() - t_open

    clock = TIMINGS.setdefault(sample_name, dict.fromkeys(("parse", "classify", "carry", "route", "count", "engine", "open", "close", "remove"), 0.0))
    clock["engine"] += t_engine
    clock["open"] += t_open

    # Four batches are in flight: while the GPU classifies batch k, a helper thread parses batch k + 1 out of the file
    # (mnc_fastq_next on all host threads) and starts its copy to the device, a second one merges batch k - 1 into the
    # carried hits (mnc_hitmap_update) and a third appends batch k - 2 to the routing folders (mnc_fastq_route, parallel
    # writes).  The C-ABI calls release the GIL; every stage handles the batches in file order, so the carried hits, the
    # appended files and the counts are what the one-batch-at-a-time loop gives.
    parsed = queue.Queue(maxsize=2)                   # (batch | None | exception)
    to_route = queue.Queue(maxsize=2)
    stop = threading.Event()

    def put(q, item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def parse_stage():
        # The first and the last batches are small: the other stages wait for the first one, and the last one's
        # classification, carry and output follow when nothing else is left to overlap them with.
        taken_reads, taken_bytes, total_bytes, k = 0, 0, reader.remaining(), 0
        try:
            while not stop.is_set():
                t0 = time.perf_counter()
                want = BATCH_READS
                k += 1
                if k <= len(BATCH_RAMP):
                    want = max(int(BATCH_READS * BATCH_RAMP[k - 1]), 1)
                elif total_bytes > 0:
                    left = reader.remaining()
                    per_read = max(taken_bytes / taken_reads, 1.0)
                    reads_left = left / per_read
                    # the tail in halves, down to an eighth of a batch -- when the GPU is what the file waits for: the last
                    # batch's carry and routing then come after everything else.  When the routing pass is (one file takes
                    # 14 GB/s from however many threads: tools/micro/page_cache_write.c), it is busy to the end whatever the
                    # batches' sizes, and small batches only add their fixed costs
                    gpu_bound = not last_index or clock["classify"] > 1.15 * clock["route"]
                    if gpu_bound and reads_left < 1.75 * BATCH_READS:
                        want = int(reads_left) + 64 if reads_left < BATCH_READS / 8 else max(int(reads_left / 2), BATCH_READS // 8)
                    elif not gpu_bound and reads_left < 1.5 * BATCH_READS:
                        want = int(reads_left) + 64                   # one last batch of up to a batch and a half
                # (bases in proportion: the parser reads ahead what a batch may hold, and the first batch is waited for)
                if not reader.next(max(want, 1), max(min(1 << 27, max(want, 1) * 6000), 1 << 20)):
                    break
                taken_reads += reader.n
                taken_bytes = total_bytes - reader.remaining() if total_bytes > 0 else 0
                batch = reader.detach()
                clock["parse"] += time.perf_counter() - t0
                _trace("parse", batch.n, t0)
                t_ann = time.perf_counter()
                try:
                    _announce(batch)
                except BaseException:
                    batch.close()
                    raise
                _trace("announce", batch.n, t_ann)
                if not put(parsed, batch):
                    batch.close()
                    return
            put(parsed, None)
        except BaseException as e:                    # a malformed file: raised where the reference would raise
            put(parsed, e)

    def _announce(batch):
        """Best effort: the batch's bases cross PCIe behind the kernels of the batch before it.  There is one spare
        buffer on the device, free again as soon as the batch announced last has STARTED its call -- so waiting makes
        sense only while the classifying thread still has an earlier batch to take out of `parsed`; with the queue
        empty the announced batch is running (the slot frees within a moment) or will never run, and the batch goes
        to the queue unannounced (its own call copies it).  A refusal for lack of device memory means the same."""
        if not batch.n:
            return
        idle_since = None
        while not stop.is_set():
            try:
                if engine.prefetch_ptr(batch.bases_ptr, batch.offsets_ptr, batch.n):
                    return
            except _capi.MncError:
                return
            if parsed.empty():
                now = time.perf_counter()
                if idle_since is None:
                    idle_since = now
                elif now - idle_since > 0.02:
                    return
            else:
                idle_since = None
            time.sleep(0.0005)

    def classify_batch(batch):
        t1 = time.perf_counter()
        try:
            out = engine.classify_ptr(batch.bases_ptr, batch.offsets_ptr, batch.n, mapping_quality)
        except _capi.MncError as err:                     # HBM exhausted beside the cached parts: free what is idle, once
            if err.code != _capi.ERR_NOMEM:
                raise
            mappy.release_idle(index.index)
            out = engine.classify_ptr(batch.bases_ptr, batch.offsets_ptr, batch.n, mapping_quality)
        clock["classify"] += time.perf_counter() - t1
        _trace("classify", batch.n, t1)
        n_skipped = int((out[0] == _capi.SKIPPED).sum())
        if n_skipped:
            # index.map() takes a read of any length (aligner.py:193, 215); the kernels stop at 2^20 bases.  Such a read
            # costs neither its sample nor its batch: it has no hits (-> unmapped/), and the run says so
            print(f"{sample_name}: {n_skipped} read(s) beyond the device limits (2^20 bases or more) not classified -> unmapped")
        return out

    def batches():
        """Parsed batches in file order; re-raises what the parser raised."""
        while True:
            item = parsed.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    paths = None
    acc = {"totals": np.zeros(0, dtype=np.int64),          # per contig name: counted amount, first read ordinal
           "first": np.zeros(0, dtype=np.int64), "seen": 0}
    decoded = []                                          # per contig name: (tax_unit, accession, in focus)
    if last_index:
        paths = [os.path.join(unmapped_folder, sample), os.path.join(ambiguous_folder, sample),
                 os.path.join(mapped_folder, sample), os.path.join(focus_folder, sample) if focus_species else None]
        for path in paths:                                # the reference opens them all in 'a' mode
            if path:
                open(path, "ab").close()

    post_error = []
    to_write = queue.Queue(maxsize=2)

    def drain(q, upstream):
        """After an error: take what the stage before still hands over, so that nobody blocks on a full queue."""
        while True:
            try:
                item = q.get(timeout=0.2)
            except queue.Empty:
                if not upstream.is_alive():
                    return
                continue
            if item is None:
                return
            item[0].close()

    def carry_stage():
        """The hits carried between index parts (`sample_hits`, aligner.py:196-203, 218-223), in file order."""
        try:
            while True:
                item = to_route.get()
                if item is None:
                    break
                batch, (assign, best, nhits) = item
                if post_error:
                    batch.close()
                    continue
                t2 = time.perf_counter()
                state = sample_hits.update(batch, index.index, assign, best, nhits)
                clock["carry"] += time.perf_counter() - t2
                _trace("carry", batch.n, t2)
                if not last_index:
                    batch.close()
                elif not put(to_write, (batch, state, sample_hits.names())):
                    batch.close()
            to_write.put(None)
        except BaseException as e:
            post_error.append(e)
            stop.set()
            to_write.put(None)
            drain(to_route, parser)

    def write_stage():
        """On the last part: routing and counting (aligner.py:232-265), in file order."""
        try:
            while True:
                item = to_write.get()
                if item is None:
                    return
                batch, state, names_now = item
                try:
                    if not post_error:
                        t3 = time.perf_counter()
                        _route_and_count(batch, state, names_now, decoded, acc, mode, overnight, focus_species, paths, clock)
                        _trace("route", batch.n, t3)
                finally:
                    batch.close()
        except BaseException as e:
            post_error.append(e)
            stop.set()
            drain(to_write, carrier)

    engine.prefetch_cancel()                           # an engine from the pool: nothing of an earlier run may be pending
    parser = threading.Thread(target=parse_stage, name="mnc-parse", daemon=True)
    carrier = threading.Thread(target=carry_stage, name="mnc-carry", daemon=True)
    writer = threading.Thread(target=write_stage, name="mnc-route", daemon=True)
    for th in (parser, carrier, writer):
        th.start()
    try:
        for batch in batches():
            if post_error:
                batch.close()
                break
            out = classify_batch(batch)
            if not put(to_route, (batch, out)):
                batch.close()
                break
    finally:
        to_route.put(None)
        carrier.join()
        writer.join()
        _trace("joined", 0, time.perf_counter())
        stop.set()
        leftover = []
        while parser.is_alive():                          # unblock a parser waiting on a full queue
            try:
                item = parsed.get(timeout=0.05)
                if item is not None and not isinstance(item, BaseException):
                    leftover.append(item)
            except queue.Empty:
                pass
        parser.join()
        while True:                                       # ... and what it had queued before it saw the stop
            try:
                item = parsed.get_nowait()
            except queue.Empty:
                break
            if item is not None and not isinstance(item, BaseException):
                leftover.append(item)
        try:
            engine.prefetch_cancel()                      # a batch announced and not classified (an error, a stop):
        except _capi.MncError:                            # its copy is over before its page-locked arrays are given back
            pass
        for item in leftover:
            item.close()
        t0 = time.perf_counter()
        reader.close()
        clock["close"] += time.perf_counter() - t0
    _trace("closed", 0, time.perf_counter())
    if post_error:
        raise post_error[0]
    if not last_index:
        sample_hits.save(carried_file)
        return None
    totals, first = acc["totals"], acc["first"]
    sample_alignment = dict()
    for u in sorted(np.flatnonzero(first >= 0), key=lambda u: first[u]):
        tax_unit, accession, _ = decoded[u]
        sample_alignment.setdefault(tax_unit, Counter()).update({accession: int(totals[u])})
    if os.path.exists(carried_file):
        os.remove(carried_file)
    print(f"{sample} done")
    t0 = time.perf_counter()
    _remove_consumed(sample)
    clock["remove"] += time.perf_counter() - t0
    # the carried hits of 100 000 reads are as many small blocks: giving them back takes 10 ms, as long as routing a batch
    # (a thread of its own, not a daemon: it finishes before the interpreter exits)
    threading.Thread(target=sample_hits.close, name="mnc-hits-free").start()
    _trace("return", 0, time.perf_counter())
    return sample_alignment, sample_name


def _remove_consumed(sample):
    """`os.remove(sample)` of the reference (aligner.py:278).  Giving a gigabyte of page cache back takes ~0.1 s, as
    long as classifying the reads in it: the file leaves the folder at once under a name monica's scan for `*fastq`
    files does not match (one rename) and is unlinked by a thread of its own (not a daemon: it finishes before the
    interpreter exits)."""
    gone = os.path.join(os.path.dirname(sample), f".{os.path.basename(sample)}.{time.time_ns():x}.consumed")
    try:
        os.rename(sample, gone)
    except OSError:
        os.remove(sample)
        return
    threading.Thread(target=_unlink_quietly, args=(gone,), name="mnc-unlink").start()


def _unlink_quietly(path):
    try:
        os.remove(path)
    except OSError:
        pass


def _route_and_count(batch, state, names, decoded, acc, mode, overnight, focus_species, paths, clock):
    """One classified batch on the last index part: its records to the routing folders (aligner.py:232-243, 265)
    and its share of the counts (aligner.py:247-263)."""
    hits, mlen, name, tied = state[:, 0], state[:, 2], state[:, 3], state[:, 4]
    mapped = (hits > 0) & (tied == 0)             # one hit, or best_hit found a unique minimum
    for ctg in names[len(decoded):]:
        decoded.append(None if ":" not in ctg else _decode(ctg, overnight, focus_species))
    used = np.unique(name[mapped])
    for u in used:
        if decoded[u] is None:
            raise IndexError("list index out of range")      # best[0].split(sep=':')[1]
    if len(acc["totals"]) < len(names):
        acc["totals"] = np.concatenate([acc["totals"], np.zeros(len(names) - len(acc["totals"]), dtype=np.int64)])
        acc["first"] = np.concatenate([acc["first"], np.full(len(names) - len(acc["first"]), -1, dtype=np.int64)])
    totals, first = acc["totals"], acc["first"]
    dest = np.where(hits == 0, _capi.TO_UNMAPPED, np.where(mapped, _capi.TO_MAPPED, _capi.TO_AMBIGUOUS)).astype(np.uint8)
    if focus_species and len(used):
        in_focus = np.array([bool(d and d[2]) for d in decoded], dtype=bool)
        dest[mapped & in_focus[np.where(mapped, name, 0)]] |= _capi.TO_FOCUS
    t0 = time.perf_counter()
    batch.route(dest, np.where(mapped, name, -1), [d[0] if d else "" for d in decoded], paths)
    clock["route"] += time.perf_counter() - t0
    t0 = time.perf_counter()
    if mode == "basic":
        amount = np.ones(batch.n, dtype=np.int64)
    elif mode == "query_length":
        amount = np.diff(batch.offsets())
    elif mode == "matching":
        amount = mlen.astype(np.int64)
    else:
        amount = None
    if amount is not None and len(used):
        np.add.at(totals, name[mapped], amount[mapped])
        ordinal = acc["seen"] + np.flatnonzero(mapped)
        for u in used:
            if first[u] < 0:
                first[u] = ordinal[np.argmax(name[mapped] == u)]
    acc["seen"] += batch.n
    clock["count"] += time.perf_counter() - t0


def _decode(ctg, overnight, focus_species):
    """Contig name -> (tax_unit or genus, accession, tax_unit in focus) (aligner.py:234-240)."""
    parts = ctg.split(sep=":")
    tax_unit, accession = parts[0], parts[1]
    in_focus = tax_unit in focus_species
    if overnight:
        tax_unit = tax_unit.split(sep="_")[0]            # tax_unit becomes the genus
    return tax_unit, accession, in_focus


def alignment_update(results, output_folder):
    """Merge per-sample counts into the persisted `alignment.pkl` (aligner.py:282-302)."""
    alignment_pickle = os.path.join(output_folder, ALIGNMENT_PICKLE_FILENAME)
    alignment = dict()
    if os.path.exists(alignment_pickle):
        with open(alignment_pickle, "rb") as f:
            alignment = pickle.load(f)
    for alignment_sample, sample_name in results:
        if sample_name not in alignment:
            alignment[sample_name] = alignment_sample
            continue
        for tax_unit, counter in alignment_sample.items():
            if tax_unit in alignment[sample_name]:
                alignment[sample_name][tax_unit].update(counter)
            else:
                alignment[sample_name][tax_unit] = counter
    with open(alignment_pickle, "wb") as f:
        pickle.dump(alignment, f)
    return alignment


def normalizer(alignment, genomes_length=None):
    """counts / genome length, then fraction of the sample total (aligner.py:305-319)."""
    if not genomes_length:
        with open(os.path.join(GENOMES_PATH, "current_genomes_length.pkl"), "rb") as f:
            genomes_length = pickle.load(f)
    for sample in alignment.keys():
        sample_total = 0
        for counter in alignment[sample].values():
            for accession, count in counter.items():
                per_base = count / genomes_length[accession]
                sample_total += per_base
                counter[accession] = per_base
        for counter in alignment[sample].values():
            for accession, per_base in counter.items():
                counter[accession] = per_base / sample_total
    return alignment


def alignment_to_data_frame(alignment, output_folder=None, filename="monica.dataframe"):
    """dict -> MultiIndex DataFrame -> CSV (aligner.py:322-325)."""
    import pandas as pd
    data_frame = pd.concat({k: pd.DataFrame(v).unstack() for k, v in alignment.items() if v}, axis=1).dropna(how="all")
    pd.DataFrame.to_csv(data_frame, os.path.join(output_folder, filename))
    return data_frame


def best_hit(hits):
    """Smallest NM/mlen wins; an exact tie for the minimum means ambiguous -> 0
    (aligner.py:328-339: the distance recorded at the last update of the minimum is zero)."""
    smallest, winner, gap = float("inf"), None, 0
    for hit in hits:
        ratio = float(hit[1]) / hit[2]
        if ratio <= smallest:
            gap = smallest - ratio
            smallest, winner = ratio, hit
    return winner if gap else 0


def any_result(alignment):
    """1 if any sample has at least one taxon (aligner.py:342-350)."""
    return 1 if any(bool(v) for v in alignment.values()) else 0
