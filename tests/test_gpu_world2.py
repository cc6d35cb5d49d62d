"""World size 2 on one GPU: two ranks as two threads, each with its own context, joined by the loopback communicator
(tests/cpp/loopback_rccl.hip — RCCL itself refuses two ranks on one device). The library binds its communicator library once per
process, so the scenario (tests/gpu_world2_loopback.py) runs in a child process with LOCGPU_RCCL_LIB set."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_give_every_rank_the_plain_batch_poses():
    stub = os.path.join(ROOT, "tests", "cpp", "libloopback_rccl.so")
    assert os.path.exists(stub), "run __graft_entry__.build() first"
    env = dict(os.environ, LOCGPU_RCCL_LIB=stub)
    env.pop("LOCGPU_SHARD_DECOUPLED", None)   # the defaults are what is under test: owner solves ahead when world > 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_world2_loopback.py")], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and "WORLD2 OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_without_the_owner_solving_ahead():
    """The same scenario with every rank waiting for the reduced sums (LOCGPU_SHARD_DECOUPLED=0: all collectives on the
    communication stream in host order)."""
    stub = os.path.join(ROOT, "tests", "cpp", "libloopback_rccl.so")
    assert os.path.exists(stub), "run __graft_entry__.build() first"
    env = dict(os.environ, LOCGPU_RCCL_LIB=stub, LOCGPU_SHARD_DECOUPLED="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_world2_loopback.py")], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0 and "WORLD2 OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
