"""Does the pinned double-buffered scan upload really hide behind the alignment? (run on the GPU box)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from loc_lib_amd import api, synth
ctx = api.Context(0)
m = synth.make_map(10_000_000); ctx.icp_set_target(m)
scans = [synth.make_scan(i) for i in range(256)]
inits = np.stack([synth.make_pose(i)[1] for i in range(256)])
b0 = ctx.batch(scans); b1 = ctx.batch(scans)
opts = api.icp_opts(method=api.P2PLANE)
for _ in range(2):
    t0 = time.perf_counter(); b1.upload_async(scans); t1 = time.perf_counter(); b1.upload_wait(); t2 = time.perf_counter()
    print("upload alone: start %.2f ms, total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
ctx.icp_align_batch(b0, inits, opts)
for _ in range(3):
    t0 = time.perf_counter(); ctx.icp_align_batch(b0, inits, opts); t1 = time.perf_counter()
    print("align alone %.2f ms" % ((t1 - t0) * 1e3))
for _ in range(3):
    t0 = time.perf_counter(); b1.upload_async(scans); ctx.icp_align_batch(b0, inits, opts); t1 = time.perf_counter(); b1.upload_wait(); t2 = time.perf_counter()
    print("align with upload %.2f ms, upload done at %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
# the bench's pattern: two batches, every step uploads the other one and aligns this one
sc = api.MarshalledScans(scans)
cases = [("alternate, no upload", False, 0), ("alternate + upload (bench pattern)", True, 0), ("alternate, no upload, HIP-event profile (all stages)", False, 1),
         ("alternate + upload, HIP-event profile (all stages)", True, 1), ("alternate, no upload, HIP-event profile (search only)", False, 2),
         ("alternate + upload, HIP-event profile (search only)", True, 2)]
for label, up, prof in cases * 3:
    ctx.profile_enable(prof)
    bufs = [b0, b1]
    for b in bufs:
        b.upload_wait()
    t0 = time.perf_counter()
    for i in range(10):
        if up:
            bufs[(i + 1) % 2].upload_async(sc)
        ctx.icp_align_batch(bufs[i % 2], inits, opts)
    for b in bufs:
        b.upload_wait()
    print("%s: %.2f ms per step" % (label, (time.perf_counter() - t0) / 10 * 1e3))
