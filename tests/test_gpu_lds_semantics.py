"""The hot search kernel keeps its DFS stack in dynamic LDS and relies on what gfx950 does with LDS addresses outside a workgroup's
allocation (loc_lib_amd/csrc/search_walk.hpp: rows below the stack's bottom must read as 0, a push above the last row must vanish,
the stack must start at LDS address 0). tests/cpp/lds_semantics.hip checks exactly that with one-wave workgroups of the kernels'
LDS sizes, thousands at once so that every CU is full of other workgroups' non-zero words. A change of hardware, driver or
compiler that breaks the assumption is named here instead of surfacing as wrong neighbour lists somewhere in the parity suite."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_out_of_allocation_lds_reads_return_zero_and_writes_are_dropped():
    exe = os.path.join(ROOT, "tests", "cpp", "lds_semantics")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "LDS SEMANTICS OK" in out.stdout, out.stdout[-3000:] + out.stderr[-1000:]
    assert out.stdout.count(": ok") == 6, out.stdout
