"""The packed-pair form of ksw2's kernel (monica_amd/csrc/ksw_pk.h: two cells per 32-bit register, tags in the low byte
of every 16-bit lane) against the oracle's literal int8 simulation of ksw_extd2_sse, on the CPU.

index.map() runs that kernel for every region (monica/genomes/aligner.py:193, 215 -> mappy 2.17 -> mm_align1); the GPU
kernel `ksw_wp` (csrc/k_align.hip) is the same arithmetic on one wave.  tests/ksw_pk_host.cpp compiles the header for the
host (the VOP3P operations emulated in plain C++) and runs the array-level form of the kernel -- ksw2's layout, its 16-lane
rounding, stale cells, tie orders, Z-drop, direction codes through kpk::decode and ksw_backtrack -- on random calls in every
mode minimap2 uses (approximate / exact maximum, extension, left- and right-aligned gaps, reversed CIGARs, bands that clip
the matrix): result fields and CIGARs must equal the oracle's."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_program(oracle, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("kswpk") / "ksw_pk_host")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "ksw_pk_host.cpp"),
                           "-L" + odir, "-lorc", "-Wl,-rpath," + odir])
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_packed_cells_equal_the_literal_simulation(host_program, seed):
    r = subprocess.run([host_program, "6000", str(seed)], stdout=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok 6000", r.stdout[-2000:]
