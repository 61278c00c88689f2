"""The host-side cache of loaded index parts (monica_amd/mappy_compat.py) without a GPU: the reference
holds one part at a time (monica/genomes/aligner.py:91-103 rebinds `index` per part); this library
keeps parts between the passes of monica's loop, bounded by count AND by bytes."""
import os

import numpy as np

from monica_amd import mappy_compat as mc, synth


def _write_parts(tmp_path, capi, n_parts):
    paths = []
    for p in range(n_parts):
        names, seqs = synth.genome_set(2, seed=100 + p, min_len=40_000, max_len=50_000, diverged_half=False)
        path = str(tmp_path / f"index{p}.mmi")
        capi.Index.from_seqs([f"G{p}_{i}:A{p}{i}.1" for i in range(2)], seqs).save(path)
        paths.append(path)
    return paths


def test_cache_is_bounded_by_bytes_and_keeps_the_newest(tmp_path, capi, monkeypatch):
    paths = _write_parts(tmp_path, capi, 5)
    size = os.path.getsize(paths[0])
    monkeypatch.setattr(mc, "_INDEX_CACHE", {})
    monkeypatch.setattr(mc, "_CACHE_BYTES", {})
    monkeypatch.setattr(mc, "_ENGINE_POOLS", {})
    monkeypatch.setattr(mc, "_INDEX_CACHE_MAX", 4)
    mc.reserve_index_cache(5)                                   # the count alone would keep all five
    assert mc._INDEX_CACHE_MAX == 5
    monkeypatch.setattr(mc, "_host_budget", lambda: int(2.5 * size))
    loaded = [mc._load_index_cached(p) for p in paths]
    assert len(mc._INDEX_CACHE) == 2                            # two parts fit the byte budget
    keys = list(mc._INDEX_CACHE)
    assert [k[0] for k in keys] == [os.path.realpath(p) for p in paths[-2:]]
    assert mc._load_index_cached(paths[-1]) is loaded[-1]       # a hit: the same object
    assert mc._load_index_cached(paths[0]) is not loaded[0]     # evicted: loaded again, as the reference would
    assert sum(mc._CACHE_BYTES.values()) <= 2.5 * size and set(mc._CACHE_BYTES) == set(mc._INDEX_CACHE)
    # one part larger than the whole budget still loads (the newest always stays)
    monkeypatch.setattr(mc, "_host_budget", lambda: 1)
    idx = mc._load_index_cached(paths[2])
    assert len(mc._INDEX_CACHE) == 1 and next(iter(mc._INDEX_CACHE.values())) is idx


def test_release_idle_drops_everything_but_the_part_in_use(tmp_path, capi, monkeypatch):
    paths = _write_parts(tmp_path, capi, 3)
    monkeypatch.setattr(mc, "_INDEX_CACHE", {})
    monkeypatch.setattr(mc, "_CACHE_BYTES", {})
    monkeypatch.setattr(mc, "_ENGINE_POOLS", {})
    monkeypatch.setattr(mc, "_INDEX_CACHE_MAX", 8)
    loaded = [mc._load_index_cached(p) for p in paths]
    assert len(mc._INDEX_CACHE) == 3

    class FakeEngine:
        closed = 0

        def close(self):
            FakeEngine.closed += 1

    def fake(index):
        e = FakeEngine()
        e.index = index
        return e

    # dropping a part closes the idle engines still bound to it (they would keep it alive), not the others
    mc._ENGINE_POOLS[0] = [fake(loaded[0]), fake(loaded[2]), fake(loaded[0])]
    with mc._INDEX_CACHE_LOCK:
        mc._drop_locked(next(iter(mc._INDEX_CACHE)))
    assert FakeEngine.closed == 2 and [e.index for e in mc._ENGINE_POOLS[0]] == [loaded[2]]
    mc._ENGINE_POOLS[0].append(fake(loaded[1]))
    mc.release_idle(keep_index=loaded[2])
    assert list(mc._INDEX_CACHE.values()) == [loaded[2]]
    assert FakeEngine.closed == 4 and all(len(p) == 0 for p in mc._ENGINE_POOLS.values())
