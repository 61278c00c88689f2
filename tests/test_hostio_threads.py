"""The FASTQ reader's pipeline of monica_amd/aligner.py (a thread parses and detaches batches, a second routes them to
mapped / unmapped / ambiguous, a third asks for the qualities of the batch in flight) on the CPU, plain and under
ThreadSanitizer: tests/hostio_tsan.cpp links csrc/hostio.cpp alone.  The routed files must hold, byte for byte, what
SeqIO.write gives for the same records (aligner.py:232-243: the mapped record's id replaced by the tax_unit)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", [[], ["-fsanitize=thread"]], ids=["plain", "tsan"])
def test_parse_route_and_qualities_side_by_side(tmp_path, flags):
    exe = str(tmp_path / "hostio_threads")
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", *flags, "-I" + os.path.join(ROOT, "include"), "-o", exe,
                            os.path.join(ROOT, "tests", "hostio_tsan.cpp"), os.path.join(ROOT, "monica_amd", "csrc", "hostio.cpp")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if build.returncode != 0 and flags:
        pytest.skip("no ThreadSanitizer runtime in this image: " + build.stdout[-200:])
    assert build.returncode == 0, build.stdout[-2000:]
    # (this image's libtsan cannot map its shadow memory under address-space randomisation: setarch -R)
    cmd = (["setarch", "x86_64", "-R"] if flags else []) + [exe, str(tmp_path), "12000"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    if flags and ("DEADLYSIGNAL" in r.stdout or "setarch:" in r.stdout):
        pytest.skip("ThreadSanitizer cannot run here: " + r.stdout[-200:])
    assert r.returncode == 0 and r.stdout.strip().endswith("ok") and "WARNING: ThreadSanitizer" not in r.stdout, r.stdout[-3000:]
