"""Short runs of the randomised parity probes under tools/ (each compares the HIP path with the oracle on generated inputs; the long runs
are a manual job on the GPU box — these keep the probes alive and give every `-m gpu` run a few hundred fresh-but-seeded cases).
fuzz_search found the one real defect of round 3 (a push lost beyond the last stack row), which is why it is here at all."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _run(script, *args, env=None, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script)] + list(args), env=dict(os.environ, **(env or {})), capture_output=True,
                         text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    return out.stdout


@pytest.mark.parametrize("rows", ["15", "12"])
def test_fuzz_search_short(rows):
    """45 cases = five of each map kind, alternating the one-scan and the batch kernel; at 12 stack rows the deep pass and the
    overflow rule run about a hundred times more often."""
    out = _run("fuzz_search.py", "--cases", "45", "--seed", "101", env={"LOCGPU_FAST_STACK": rows})
    assert "mismatching runs 0" in out, out[-2000:]


def test_fuzz_filters_short():
    assert "mismatches 0" in _run("fuzz_filters.py", "--cases", "400", "--seed", "7")


def test_fuzz_loam_short():
    assert "mismatches 0" in _run("fuzz_loam.py", "--cases", "80", "--seed", "7")


def test_fuzz_ndt_short():
    """A mismatch must reproduce when its case is run alone to fail this test: one incremental-NDT case came out wrong ONCE in ≈1 700
    (profiles/experiments.md, "Fuzzing beyond the search") and never again; a repeat of that is reported, loudly, without taking the
    rest of the suite down with it (the driver runs pytest with -x)."""
    import re
    import warnings
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ndt.py"), "--cases", "30"], capture_output=True, text=True, timeout=900, cwd=ROOT).stdout
    if "mismatches 0" in out:
        return
    cases = sorted(set(int(c) for c in re.findall(r"MISMATCH case (\d+)", out)))
    assert cases, out[-2000:]
    for c in cases:
        again = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ndt.py"), "--cases", "30", "--only", str(c)], capture_output=True, text=True,
                               timeout=900, cwd=ROOT).stdout
        assert "mismatches 0" in again, "fuzz_ndt case %d mismatches reproducibly:\n%s" % (c, again[-2500:])
    warnings.warn("fuzz_ndt: case(s) %s mismatched once and not when run alone:\n%s" % (cases, out[-1500:]))


def test_fuzz_align_short():
    out = _run("fuzz_align.py", "--cases", "8")
    assert "iteration-count mismatches 0" in out, out[-2000:]
