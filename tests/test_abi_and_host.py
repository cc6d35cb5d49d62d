"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/locgpu.h declares, fails
loudly without a GPU, and its host logic (packed KD-tree ingest, Gauss–Newton update) matches the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "locgpu.h")).read()
    return sorted(set(re.findall(r"LOCGPU_API\s+[\w\s\*]+?\b(locgpu_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol(api):
    lib = api.lib()
    declared = _header_functions()
    assert len(declared) >= 25
    missing = [f for f in declared if not hasattr(lib, f)]
    assert not missing, missing
    assert sorted(api.ABI_SYMBOLS) == declared  # the Python binding covers the whole header, nothing more


def test_header_cites_reference_for_every_entry_point():
    text = open(os.path.join(ROOT, "include", "locgpu.h")).read()
    for fn in ("locgpu_icp_set_target", "locgpu_knn", "locgpu_icp_hb", "locgpu_icp_align", "locgpu_transform_cloud",
               "locgpu_ndt_set_target", "locgpu_ndt_align"):
        pos = text.index(fn + "(")
        block = text[max(0, pos - 900):pos]
        assert re.search(r"\.(cpp|h|hpp):\d+", block), fn


def test_option_defaults_are_the_reference_defaults(api):
    o = api.icp_opts()
    # IcpOptions, icp_registration.hpp:29-38
    assert (o.method, o.max_iteration, o.max_nn_distance, o.max_plane_distance, o.max_line_distance, o.min_effective_pts, o.eps) == \
        (0, 20, 1.0, 0.1, 0.5, 10, 1e-2)
    assert o.approximate == 1 and abs(o.ann_alpha - 0.1) < 1e-7  # kdtree.h:128-129
    n = api.ndt_opts()
    # NdtOptions, ndt_registration.hpp:28-41
    assert (n.max_iteration, n.voxel_size, n.min_effective_pts, n.min_pts_in_voxel, n.eps, n.res_outlier_th, n.nearby_type) == \
        (20, 1.0, 10, 3, 1e-2, 20.0, 1)


def test_no_gpu_means_loud_failure_not_cpu_fallback(api):
    if api.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(api.LocGpuError) as e:
        api.Context(0)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under loc_lib_amd/ or include/ may reference it."""
    bad = []
    for base in ("loc_lib_amd", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"\blocref\b|oracle[/\.]", txt):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_no_null_stream_fills_in_the_library():
    """Round 4, found by `tools/fuzz_align.py --cases 120` (about one first call in a thousand): the batch allocation zero-filled the scan
    counts with hipMemset — NULL stream, not complete when it returns — while the context's streams are non-blocking, i.e. not ordered
    behind it: the fill could land BEHIND the first upload of the counts, and the first alignment of a fresh one-scan batch then ran its
    20 iterations on a scan of zero points and handed the initial pose back (the same call repeated was right). The race cannot be forced
    from outside (a busy NULL stream holds the allocation's own hipMalloc calls back), so the rule is pinned in the sources: device fills
    go through `fill_now` / hipMemsetAsync on a stream of the context, kernels are never launched on the default stream."""
    bad = []
    csrc = os.path.join(ROOT, "loc_lib_amd", "csrc")
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".hpp", ".cpp")):
            continue
        for ln, line in enumerate(open(os.path.join(csrc, fn), errors="ignore"), 1):
            code = line.split("//")[0]
            if re.search(r"\bhipMemset\s*\(", code) or re.search(r"\bhipMemsetD\d+\s*\(", code) or "<<<" in code:
                bad.append("%s:%d: %s" % (fn, ln, line.strip()[:120]))
            if re.search(r"hipLaunchKernelGGL\([^;]*,\s*0,\s*(0|nullptr|NULL)\s*,", code):
                bad.append("%s:%d: default-stream launch: %s" % (fn, ln, line.strip()[:120]))
    assert not bad, bad


# ---------------------------------------------------------------------------------------------- host logic: tree ingest
def _packed_tree(api, xyz):
    L = api.lib()
    L.locgpu_debug_build_tree.restype = ctypes.c_size_t
    L.locgpu_debug_build_tree.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    xyz = np.ascontiguousarray(xyz[:, :3], dtype=np.float32)
    info = np.zeros(3, dtype=np.int64)
    n = L.locgpu_debug_build_tree(xyz.ctypes.data, len(xyz), None, 0, info.ctypes.data)
    slots = np.zeros(n, dtype=np.uint64)
    L.locgpu_debug_build_tree(xyz.ctypes.data, len(xyz), slots.ctypes.data, n, info.ctypes.data)
    return slots, info


def _preorder(slots):
    lo = (slots & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (slots >> np.uint64(32)).astype(np.uint32)
    axis, th, pidx, right = [], [], [], []
    i = 0
    while i < len(slots):
        tag = int(hi[i] >> 30)
        if tag == 3:
            axis.append(-1); th.append(0.0); pidx.append(int(hi[i] & 0x3FFFFFFF)); right.append(-1)
            i += 2
        else:
            axis.append(tag); th.append(float(lo[i:i + 1].view(np.float32)[0])); pidx.append(-1); right.append(int(hi[i] & 0x3FFFFFFF))
            i += 1
    return np.array(axis, np.int32), np.array(th, np.float32), np.array(pidx, np.int32), right


@pytest.mark.parametrize("case", ["uniform", "city", "duplicates", "tiny", "planar"])
def test_packed_tree_equals_oracle_tree(api, locref, synth, case):
    rng = np.random.RandomState(7)
    if case == "uniform":
        pts = (rng.rand(30000, 3) * 50).astype(np.float32)
    elif case == "city":
        pts = synth.make_map(150000)
    elif case == "duplicates":
        pts = (rng.rand(2000, 3) * 5).astype(np.float32)
        pts[100:400] = pts[100]
        pts[1000:1010, 0] = pts[1000, 0]
    elif case == "tiny":
        pts = (rng.rand(3, 3)).astype(np.float32)
    else:
        pts = np.c_[rng.rand(20000, 2) * 30, np.zeros(20000)].astype(np.float32)  # zero variance on z
    slots, info = _packed_tree(api, pts)
    tree = locref.KdTree(pts)
    assert tuple(int(v) for v in info) == (tree.num_leaves, tree.num_nodes, tree.depth)
    a, t, p = tree.dump()
    a2, t2, p2, _ = _preorder(slots)
    np.testing.assert_array_equal(a, a2)
    np.testing.assert_array_equal(t.view(np.uint32), t2.view(np.uint32))  # thresholds bit-identical (f32 sequential sums)
    np.testing.assert_array_equal(p, p2)


def test_packed_tree_equals_oracle_tree_randomised(api, locref):
    """Round 4 rewrote the host build (contiguous point records, SSE lanes, a child's sum taken in its parent's partition pass, no level
    barriers: positions are laid out for 3m − 1 slots per m-point sub-tree, duplicates fall back to the level-by-level path). Forty
    seeded clouds — 1 … 60 000 points; uniform, planar, clustered, on a lattice (ties on every axis), a few to most of the points
    duplicated, all points equal, one coordinate constant, huge and tiny magnitudes — must give the oracle's tree: axes, point order,
    counts, depth, and split thresholds BIT for bit."""
    rng = np.random.RandomState(20260)
    sizes = [1, 2, 3, 4, 5, 7, 16, 33, 100, 511, 1024, 2049, 5000, 20000, 60000]
    kinds = ["uniform", "planar", "clusters", "lattice", "dups_few", "dups_most", "all_equal", "const_axis", "huge", "tiny"]
    for case in range(40):
        n = sizes[case % len(sizes)]
        kind = kinds[(case * 7 + case // len(sizes)) % len(kinds)]
        pts = (rng.rand(n, 3) * 50 - 25).astype(np.float32)
        if kind == "planar":
            pts[:, 2] = (0.02 * rng.randn(n)).astype(np.float32)
        elif kind == "clusters":
            c = (rng.rand(max(1, n // 200), 3) * 80).astype(np.float32)
            pts = (c[rng.randint(0, len(c), n)] + 0.05 * rng.randn(n, 3)).astype(np.float32)
        elif kind == "lattice":
            pts = rng.randint(0, 6, size=(n, 3)).astype(np.float32)
        elif kind == "dups_few" and n > 3:
            pts[n // 3: n // 3 + max(2, n // 20)] = pts[n // 3]
        elif kind == "dups_most" and n > 3:
            pts = pts[rng.randint(0, max(2, n // 50), n)]
        elif kind == "all_equal":
            pts[:] = pts[0]
        elif kind == "const_axis":
            pts[:, rng.randint(0, 3)] = np.float32(3.25)
        elif kind == "huge":
            pts *= np.float32(1e15)
        elif kind == "tiny":
            pts *= np.float32(1e-20)
        slots, info = _packed_tree(api, pts)
        tree = locref.KdTree(pts)
        assert tuple(int(v) for v in info) == (tree.num_leaves, tree.num_nodes, tree.depth), (case, kind, n)
        a, t, p = tree.dump()
        a2, t2, p2, _ = _preorder(slots)
        assert np.array_equal(a, a2), (case, kind, n)
        assert np.array_equal(t.view(np.uint32), t2.view(np.uint32)), (case, kind, n)
        assert np.array_equal(p, p2), (case, kind, n)


def test_packed_tree_child_links_reproduce_knn(api, locref):
    """Walk the packed slots (left = slot+1, right = stored index) with the reference's DFS in pure Python (small case)."""
    rng = np.random.RandomState(8)
    pts = (rng.rand(600, 3) * 10).astype(np.float32)
    slots, _ = _packed_tree(api, pts)
    lo = (slots & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (slots >> np.uint64(32)).astype(np.uint32)
    f = lambda u: np.array([u], dtype=np.uint32).view(np.float32)[0]
    tree = locref.KdTree(pts)
    q = (rng.rand(40, 3) * 10).astype(np.float32)
    ref = tree.knn(q, k=5, approximate=True, alpha=0.1)
    import heapq
    for qi in range(len(q)):
        heap = []  # max-heap via negated distance; ties are irrelevant on random data

        def knn(slot):
            tag = int(hi[slot] >> 30)
            if tag == 3:
                p = np.array([f(lo[slot]), f(lo[slot + 1]), f(hi[slot + 1])], dtype=np.float32)
                d = q[qi] - p
                d2 = np.float32(d[0] * d[0]) + (np.float32(d[1] * d[1]) + np.float32(d[2] * d[2]))
                idx = int(hi[slot] & 0x3FFFFFFF)
                if len(heap) < 5:
                    heapq.heappush(heap, (-d2, idx))
                elif d2 < -heap[0][0]:
                    heapq.heapreplace(heap, (-d2, idx))
                return
            th = f(lo[slot])
            left, right = slot + 1, int(hi[slot] & 0x3FFFFFFF)
            this, that = (left, right) if q[qi][tag] < th else (right, left)
            knn(this)
            dd = np.float32(q[qi][tag] - th)
            if len(heap) < 5 or np.float32(dd * dd) < np.float32(np.float32(-heap[0][0]) * np.float32(0.1)):
                knn(that)

        knn(0)
        got = [i for _, i in sorted(heap, key=lambda t: -t[0])]
        assert got == list(ref[qi])


# ---------------------------------------------------------------------------------------------- host logic: GN update
def test_gn_update_matches_oracle(api, locref):
    rng = np.random.RandomState(9)
    for method in (0, 2):
        J = rng.randn(50, 6)
        H = J.T @ J
        B = rng.randn(6) * 0.1
        hb = np.concatenate([H.reshape(-1), B, [50.0, 1.0]])
        pose = np.array([0.1, -0.2, 0.3, 0.9, 1.0, 2.0, 3.0])
        pose[:4] /= np.linalg.norm(pose[:4])
        new_pose, dx, applied, stop = api.gn_update(hb, method, 10, 1e-2, pose)
        det, x = locref.lu6(H, B)
        if method == 0:
            x = x / 16  # icp cpp:287
        np.testing.assert_allclose(dx, x, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(new_pose, locref.apply_update(pose, x), rtol=0, atol=1e-15)
        assert applied and stop == (np.linalg.norm(x) < 1e-2)
    # too few effective points, or singular H: no update (icp cpp:204-211)
    hb = np.concatenate([np.eye(6).reshape(-1), np.ones(6), [5.0, 0.0]])
    p2, dx, applied, stop = api.gn_update(hb, 2, 10, 1e-2, pose)
    assert not applied and not stop and np.array_equal(p2, pose)
    hb = np.concatenate([np.zeros(36), np.ones(6), [50.0, 0.0]])
    p2, dx, applied, stop = api.gn_update(hb, 2, 10, 1e-2, pose)
    assert not applied and np.array_equal(p2, pose)


def test_host_ingest_is_clean_under_asan_ubsan(tmp_path):
    """The multithreaded host tree builder (csrc/kdtree_build.cpp) under AddressSanitizer + UBSan (CPU build only)."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "loc_lib_amd", "csrc")
    exe = str(tmp_path / "host_build_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-I", csrc,
           os.path.join(ROOT, "tests", "cpp", "host_build_sanitize.cpp"), os.path.join(csrc, "kdtree_build.cpp"),
           "-o", exe, "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan harness ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_host_tree_build_pool_under_thread_sanitizer(tmp_path):
    """the persistent pool of the host tree build (kdtree_build.cpp) with four concurrent callers — what two
    contexts ingesting at once amount to — under ThreadSanitizer, every result equal to the single-caller build; then a forked
    child builds with a pool of its own (the parent's threads do not exist there). GPU sanitizers do not exist on this
    pool, the host side is where the threads are."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "loc_lib_amd", "csrc")
    exe = str(tmp_path / "host_build_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-I", csrc,
           os.path.join(ROOT, "tests", "cpp", "host_build_tsan.cpp"), os.path.join(csrc, "kdtree_build.cpp"), "-o", exe, "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and ("sanitize" in r.stderr or "tsan" in r.stderr):
        pytest.skip("ThreadSanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1 die_after_fork=0", LOCGPU_BUILD_THREADS="6")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert r.returncode == 0 and "tsan harness ok" in r.stdout, r.stdout[-1000:] + r.stderr[-4000:]


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_inc_ndt_host_replay_under_sanitizers(tmp_path, san):
    """the host side of SetIncNdtTargetCloud (csrc/inc_ndt_lru.hpp — the per-point LRU replay used when a
    cloud's working set exceeds the voxel capacity) under ASan + UBSan and under TSan, checked against a plain restatement of the
    reference's list + map loop (ndt_registration.cpp:150-171) on random multi-call sequences; the same harness proves the closed
    form the device path uses (top capacity-1 recency stamps) equal to the sequential replay whenever it applies."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "loc_lib_amd", "csrc")
    exe = str(tmp_path / "inc_ndt_host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-I", csrc, os.path.join(ROOT, "tests", "cpp", "inc_ndt_host_sanitize.cpp"), "-o", exe]
    if san != "thread":
        cmd.insert(5, "-fno-sanitize-recover=undefined")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and ("sanitize" in r.stderr or "tsan" in r.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    if r.returncode != 0 and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert r.returncode == 0 and "inc-ndt host harness ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_pool_bookkeeping_is_lock_step_across_ranks_under_sanitizers(tmp_path):
    """The N > 1 path of the open-scan pool on the CPU: csrc/pool_sched.hpp — the slots, the source regions, the FIFO
    of waiting jobs: everything scan_pool.hip decides on the host — driven as 2 and as 8 simulated ranks that each hold a different
    shard of every job. Every rank must give every scan the same slot and region at the same chunk boundary and issue the same
    number of collectives (RCCL would hang, or sum the wrong rows, otherwise); one rank with copies that arrive late must keep FIFO
    order and lose no slot or region. ASan + UBSan on."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "loc_lib_amd", "csrc")
    exe = str(tmp_path / "pool_sched_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-I", csrc,
           os.path.join(ROOT, "tests", "cpp", "pool_sched_sanitize.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    for args in (["ranks", "2"], ["ranks", "8"], ["stall"]):
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.count(" OK") >= 3, r.stdout[-1000:] + r.stderr[-3000:]


def test_finished_scan_post_rejects_torn_reads(tmp_path):
    """The finished-scan post of a one-scan alignment paced from the host (csrc/gn_post.hpp): the solve kernel's stores to pinned host
    memory carry no fence, so the record and its {tag, checksum} pair may arrive in any order. tests/cpp/gn_post_sanitize.cpp plays the
    device: it makes a new post visible ONE WORD AT A TIME — random orders, the pair overtaking the record, the checksum before the tag —
    over the previous post's words and asks gn_post_take() after every word: never a yes before every word it hands back is the new
    post's, never a yes for a stale call number or an iteration already seen, always a yes once everything is there; then a writer
    thread against a polling reader for 20 000 posts. ASan + UBSan on. (What's weak 9)"""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "gn_post_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-pthread",
           "-I", os.path.join(ROOT, "loc_lib_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "gn_post_sanitize.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "accepted early 0, wrong content 0, missed 0" in r.stdout and "20000 taken, 0 wrong" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


def test_context_helper_thread_under_tsan(tmp_path):
    """The context's helper thread (csrc/host_worker.hpp: the output cloud of locgpu_*_scan_match is sized, field-copied and half of
    its coordinates written there, beside the caller's thread) under ThreadSanitizer: results handed back through wait(), a run() behind
    a busy job, jobs that capture the caller's stack, destruction while busy / idle / never started (tests/cpp/host_worker_tsan.cpp)."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "host_worker_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-pthread", "-I", os.path.join(ROOT, "loc_lib_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "host_worker_tsan.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "host_worker: 0 problems" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stdout[-1000:] + r.stderr[-3000:]


# ----------------------------------------------------------------------------------------------- bench.py launcher (no GPU needed)
def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_launcher_command_and_environment():
    """`python bench.py --gpus N` without WORLD_SIZE starts N ranks itself (one process per GPU): the job's argv is
    torch.distributed.run on 127.0.0.1 with this script and its own arguments; the rank environment is the caller's without the
    rank variables of an enclosing job and with dmabuf IPC on."""
    b = _load_bench()
    cmd = b.launcher_command(8, 29511, ["--gpus", "8", "--steps", "20", "--warmup", "5"], python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    env = b.launcher_env({"PATH": "/usr/bin", "RANK": "3", "WORLD_SIZE": "4", "LOCAL_RANK": "3", "MASTER_PORT": "1", "FOO": "bar"})
    assert "RANK" not in env and "WORLD_SIZE" not in env and "LOCAL_RANK" not in env and "MASTER_PORT" not in env
    assert env["FOO"] == "bar" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["LOCGPU_BENCH_SELF_LAUNCHED"] == "1"
    assert b.launcher_env({"HSA_ENABLE_IPC_MODE_LEGACY": "1"})["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"  # an explicit setting is kept


def test_bench_refuses_more_gpus_than_the_node_has():
    """--gpus N above the node's GPU count is an error, never a silent run on fewer ranks (this container has no GPU at all), and a
    WORLD_SIZE that disagrees with --gpus is refused too."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4096", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "--gpus 4096 but this node has" in out.stderr, out.stderr[-800:]
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=300, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in out.stderr, out.stderr[-800:]


def test_facade_refuses_the_option_branches_that_are_not_on_the_gpu_path():
    """IcpMethod::PCLICP (icp_registration.cpp:385-399), NdtMethod::PCL_NDT (ndt_registration.cpp:69,246),
    IcpOptions::use_initial_translation_ = false (:273,311,351) and NdtOptions::remove_centroid_ = true (:380-384) used to be ignored by
    the façade. Now SetInputTarget and ScanMatch return false, LastError() says why, and neither the pose nor the output cloud is
    touched. A refusal needs no GPU (it comes before a context exists), so this runs in the CPU suite."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "facade_scanmatch")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "loc_lib_amd", "host"), "../../tests/cpp/facade_scanmatch"], stdout=subprocess.DEVNULL)
    r = subprocess.run([exe, "refusals"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    for what in ("PCLICP", "use_initial_translation_", "PCL_NDT", "remove_centroid_"):
        assert what + ": SetInputTarget 0 ScanMatch 0 outputs untouched 1" in r.stdout, r.stdout


def test_host_tree_build_without_sse_builds_the_same_trees(tmp_path):
    """kdtree_build.cpp used SSE intrinsics unconditionally. The lane-by-lane fallback (what an aarch64 host compiles;
    forced here with -DLOCGPU_SCALAR_VEC) must build bit-identical trees: forty clouds, one checksum over all their slots."""
    import subprocess
    csrc = os.path.join(ROOT, "loc_lib_amd", "csrc")
    sums = []
    for tag, extra in (("sse", []), ("scalar", ["-DLOCGPU_SCALAR_VEC"])):
        exe = str(tmp_path / ("host_build_" + tag))
        cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I", csrc] + extra + [os.path.join(ROOT, "tests", "cpp", "host_build_sanitize.cpp"),
               os.path.join(csrc, "kdtree_build.cpp"), "-o", exe, "-pthread"]
        subprocess.check_call(cmd)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "asan harness ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
        sums.append([ln for ln in r.stdout.splitlines() if ln.startswith("tree checksum")][0])
    assert sums[0] == sums[1], sums
