"""csrc/team.h -- the helper threads of the FASTQ reader's parse and routing passes (they sleep between passes: the GPU
box grants 16 cores of 256 and an OpenMP runtime's idle helpers spun through a third of them) -- on the CPU, plain and
under ThreadSanitizer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", [[], ["-fsanitize=thread", "-g"]], ids=["plain", "tsan"])
def test_team_passes(tmp_path, flags):
    exe = str(tmp_path / "team_host")
    build = subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", *flags, "-o", exe, os.path.join(ROOT, "tests", "team_host.cpp")],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if build.returncode != 0 and flags:
        pytest.skip("no ThreadSanitizer runtime in this image: " + build.stdout[-200:])
    assert build.returncode == 0, build.stdout[-2000:]
    # (this image's libtsan cannot map its shadow memory under address-space randomisation: setarch -R)
    cmd = ["setarch", "x86_64", "-R", exe] if flags else [exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    if flags and ("DEADLYSIGNAL" in r.stdout or "setarch" in r.stdout):
        pytest.skip("ThreadSanitizer cannot run here: " + r.stdout[-200:])
    assert r.returncode == 0 and r.stdout.strip().endswith("ok") and "WARNING: ThreadSanitizer" not in r.stdout, r.stdout[-3000:]
