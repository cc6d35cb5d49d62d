// loc_lib_amd/csrc/batch_upload.hip — see batch_upload.hpp.
#include "batch_upload.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include "context.hpp"

namespace locgpu {

// strided points → float4 {x, y, z, 0} (declared in batch_upload.hpp).
void pack_points(const char* base, size_t stride, size_t n, float4* dst) {
    if (stride == sizeof(float4)) {  // already {x, y, z, w}: no kernel of the matcher reads w
        std::memcpy(dst, base, n * sizeof(float4));
        return;
    }
    size_t i = 0;
#if defined(__SSE2__)
    if (stride >= 12 && n > 1) {
        // one unaligned 16-byte load per point (x, y, z and four bytes of whatever follows: the last point is left to the scalar
        // loop below so that nothing is read past the cloud), w masked to zero, one 16-byte store: ≈4× the scalar loop, and the
        // packing — 29.5 M points per 256-scan step — is what bounds the streaming rate with two alignments in flight
        const __m128 mask = _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1));
        for (; i + 1 < n; ++i) _mm_storeu_ps(reinterpret_cast<float*>(dst + i), _mm_and_ps(_mm_loadu_ps(reinterpret_cast<const float*>(base + i * stride)), mask));
    }
#endif
    for (; i < n; ++i) {
        float v[3];
        std::memcpy(v, base + i * stride, 12);
        dst[i] = float4{v[0], v[1], v[2], 0.f};
    }
}

namespace {

struct Unit { int scan; size_t off, len; };

void release_resources(Uploader& u) {
    if (u.stream) (void)hipStreamSynchronize(u.stream);
    for (float4* p : u.h_slots) if (p) (void)hipHostFree(p);
    for (hipEvent_t e : u.slot_ev) if (e) (void)hipEventDestroy(e);
    u.h_slots.clear(); u.slot_ev.clear(); u.slot_busy.clear();
}

// At least `want_slots` pinned slots (capped by the thread count). All or nothing: on a failure everything that was built is
// released again, so that the next request starts from scratch instead of finding a half-initialised uploader. Service thread only.
bool ensure_slots(Uploader& u, size_t want_slots, std::string& err) {
    const size_t cap = (size_t)u.n_threads * Uploader::kSlotsPerThread;
    const size_t want = std::min(cap, std::max<size_t>(1, want_slots));
    while (u.h_slots.size() < want) {
        float4* p = nullptr;
        hipEvent_t ev = nullptr;
        const hipError_t e1 = hipHostMalloc((void**)&p, Uploader::kSlotPoints * sizeof(float4));
        const hipError_t e2 = e1 == hipSuccess ? hipEventCreateWithFlags(&ev, hipEventDisableTiming) : e1;
        if (e2 != hipSuccess) {
            err = std::string("upload: pinned staging: ") + hipGetErrorString(e2);
            if (p) (void)hipHostFree(p);
            release_resources(u);
            return false;
        }
        u.h_slots.push_back(p);
        u.slot_ev.push_back(ev);
        u.slot_busy.push_back(0);
    }
    return true;
}

// One request, on the service thread: pack into pinned slots with a few packer threads, stream the slots to HBM on the copy stream,
// record the request's event behind the last copy. The status goes to the request's state.
void run_upload(locgpu_ctx* ctx, Uploader& u, UploadRequest& rq) {
    locgpu_batch* b = rq.b;
    BatchUploadState& bst = *rq.st;
    static const bool dbg = getenv("LOCGPU_UPLOAD_DEBUG") != nullptr;
    const auto t_entry = std::chrono::steady_clock::now();
    int rc = LOCGPU_OK;
    std::string err;
    // work units: every scan in slot-sized pieces
    std::vector<Unit> units;
    const int n_up = (int)rq.counts.size();
    const bool to_slots = !rq.dst.empty();
    for (int s = 0; s < n_up; ++s)
        for (size_t o = 0; o < rq.counts[s]; o += Uploader::kSlotPoints) units.push_back({s, o, std::min(Uploader::kSlotPoints, rq.counts[s] - o)});
    if (!ensure_slots(u, units.size(), err)) rc = LOCGPU_ERR_OOM;
    {   // test hook (tests/test_gpu_pool.py): LOCGPU_TEST_FAIL_UPLOAD=n makes the n-th copy into a pool's regions of this process fail
        static const int fail_at = [] { const char* e = getenv("LOCGPU_TEST_FAIL_UPLOAD"); return e ? atoi(e) : 0; }();
        static std::atomic<int> seen{0};
        if (fail_at > 0 && to_slots && ++seen == fail_at) { rc = LOCGPU_ERR_NO_DEVICE; err = "batch upload: injected failure (LOCGPU_TEST_FAIL_UPLOAD)"; }
    }
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    std::atomic<long long> wait_us{0}, pack_us{0};
    const int n_slots = (int)u.h_slots.size();
    const int nt = (int)std::min<size_t>(std::min<size_t>((size_t)u.n_threads, (size_t)std::max(1, n_slots)), std::max<size_t>(1, units.size()));
    auto packer = [&](int t) {
        (void)hipSetDevice(ctx->device);
        // thread t owns slots t, t + nt, …
        int slot = t;
        for (size_t i = next.fetch_add(1); i < units.size() && !failed.load(); i = next.fetch_add(1)) {
            const Unit& w = units[i];
            // the slot's previous copy — of this upload or of an earlier one — has left it
            const auto ta = std::chrono::steady_clock::now();
            if (u.slot_busy[slot] && hipEventSynchronize(u.slot_ev[slot]) != hipSuccess) { failed = 1; break; }
            const auto tb = std::chrono::steady_clock::now();
            pack_points((const char*)rq.srcs[w.scan] + w.off * rq.stride, rq.stride, w.len, u.h_slots[slot]);
            if (dbg) {
                wait_us += std::chrono::duration_cast<std::chrono::microseconds>(tb - ta).count();
                pack_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tb).count();
            }
            float4* base = to_slots ? rq.dst_base + (size_t)rq.dst[w.scan] * b->max_n : b->d_src + (size_t)w.scan * b->max_n;
            if (hipMemcpyAsync(base + w.off, u.h_slots[slot], w.len * sizeof(float4), hipMemcpyHostToDevice, u.stream) != hipSuccess ||
                hipEventRecord(u.slot_ev[slot], u.stream) != hipSuccess) { failed = 1; break; }
            u.slot_busy[slot] = 1;
            slot = slot + nt < n_slots ? slot + nt : t;
        }
    };
    if (rc == LOCGPU_OK) {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(packer, t);
        packer(0);
        for (auto& t : th) t.join();
        bool ok = !failed.load();
        if (!to_slots) {
            for (int s = 0; s < n_up; ++s) bst.h_counts[s] = (int)rq.counts[s];
            ok = ok && hipMemcpyAsync(b->d_counts, bst.h_counts, (size_t)n_up * sizeof(int), hipMemcpyHostToDevice, u.stream) == hipSuccess;
        }
        ok = ok && hipEventRecord(bst.done, u.stream) == hipSuccess;
        if (!ok) { rc = LOCGPU_ERR_NO_DEVICE; err = std::string("batch upload: ") + hipGetErrorString(hipGetLastError()); }
    }
    if (dbg)
        fprintf(stderr, "[upload] %d threads, %zu pieces: queued %.2f ms, %.2f ms wall, per thread %.2f ms waiting for slots + %.2f ms packing\n", nt, units.size(),
                std::chrono::duration<double, std::milli>(t_entry - rq.t_start).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count(), wait_us.load() / 1e3 / nt, pack_us.load() / 1e3 / nt);
    // the failure stays with the batch / job it happened to: its source array is partly copied and its `done` event was not recorded
    // again, so every later use (align, upload_wait, pool admission) must fail until a new upload replaces the scans
    bst.rc = rc;
    bst.err = err;
    if (rc != LOCGPU_OK) bst.done_valid = false;
}

void service_loop(locgpu_ctx* ctx) {
    Uploader& u = *ctx->up;
    (void)hipSetDevice(ctx->device);
    std::unique_lock<std::mutex> lk(u.m);
    for (;;) {
        u.cv_work.wait(lk, [&] { return u.stop || !u.queue.empty(); });
        if (u.queue.empty()) break;  // stop, and nothing left
        UploadRequest rq = std::move(u.queue.front());
        u.queue.pop_front();
        u.busy = true;
        lk.unlock();
        run_upload(ctx, u, rq);
        rq.st->pending.fetch_sub(1, std::memory_order_release);
        lk.lock();
        u.busy = false;
        u.cv_idle.notify_all();
    }
}

// Common tail of the two entry points: the request joins the context's FIFO; `st` receives the event and the status.
int enqueue(locgpu_batch* b, BatchUploadState& st, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n, const int* dst, float4* dst_base) {
    locgpu_ctx* ctx = b->ctx;
    if (!ctx->up) {
        ctx->up = new Uploader();
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        ctx->up->n_threads = (int)std::min<unsigned>(8u, std::max(2u, hw / 4));
        ctx->up->stream = ctx->copy_stream;  // created with the context (see locgpu_create)
    }
    Uploader& u = *ctx->up;
    // One upload of a batch (or job) at a time, end to end: the previous one's host side is through and its copies (same
    // destination, same pinned counts) have landed.
    (void)upload_join_state(ctx, &st);
    if (!st.done && !hip_ok(ctx, hipEventCreateWithFlags(&st.done, hipEventDisableTiming), "upload: hipEventCreate")) { st.done = nullptr; return LOCGPU_ERR_OOM; }
    if (!dst && !st.h_counts && !hip_ok(ctx, hipHostMalloc((void**)&st.h_counts, (size_t)b->n_scans * sizeof(int)), "upload: hipHostMalloc counts")) { st.h_counts = nullptr; return LOCGPU_ERR_OOM; }
    if (st.done_valid && !hip_ok(ctx, hipEventSynchronize(st.done), "batch upload: previous upload")) return LOCGPU_ERR_NO_DEVICE;
    st.rc = LOCGPU_OK;
    st.err.clear();
    st.done_valid = true;
    UploadRequest rq;
    rq.b = b;
    rq.st = &st;
    rq.srcs.assign(srcs, srcs + n);
    rq.counts.assign(counts, counts + n);
    if (dst) rq.dst.assign(dst, dst + n);
    rq.dst_base = dst_base;
    rq.stride = stride_bytes;
    rq.t_start = std::chrono::steady_clock::now();
    st.pending.fetch_add(1, std::memory_order_acq_rel);
    {
        std::lock_guard<std::mutex> lk(u.m);
        u.queue.push_back(std::move(rq));
        if (!u.started) { u.started = true; u.worker = std::thread(service_loop, ctx); }
    }
    u.cv_work.notify_one();
    return LOCGPU_OK;
}

}  // namespace

int upload_start(locgpu_batch* b, const void* const* srcs, const size_t* counts, size_t stride_bytes) {
    locgpu_ctx* ctx = b->ctx;
    if (!srcs || !counts || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: bad arguments");
    for (int s = 0; s < b->n_scans; ++s) {
        if (counts[s] > (size_t)b->max_n) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: a scan has more points than the batch was created for");
        if (counts[s] && !srcs[s]) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: NULL scan pointer");
    }
    // One alignment per batch at a time, and no upload into a batch whose alignment has been begun and not ended: the
    // chunks locgpu_align_batch_end enqueues later read d_src / d_counts / b->counts, which this upload would replace under them.
    // A caller with several alignments in flight rotates depth + 1 batches (bench.py) — the copy always goes to an idle one.
    // No ordering behind the batch's compute stream is needed beyond that: an ended alignment leaves its stream synchronised
    // (align_finish synchronises it on its error paths as well) — or, a paced one-scan alignment, with at most a few launches
    // queued that return on the scan's `done` flag before they read the counts or the points (and the copies are ordered behind
    // them all the same, below).
    if (b->pending.active) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: an alignment of this batch has been begun and not finished");
    if (b->paced_tail) {
        // belt and braces for the idle launches a paced alignment may have left queued: the copies go behind them
        b->paced_tail = false;
        if (!b->tail_ev && !hip_ok(ctx, hipEventCreateWithFlags(&b->tail_ev, hipEventDisableTiming), "batch upload: hipEventCreate")) return LOCGPU_ERR_NO_DEVICE;
        if (!hip_ok(ctx, hipEventRecord(b->tail_ev, b->stream), "batch upload: hipEventRecord") ||
            !hip_ok(ctx, hipStreamWaitEvent(ctx->copy_stream, b->tail_ev, 0), "batch upload: hipStreamWaitEvent")) return LOCGPU_ERR_NO_DEVICE;
    }
    const int rc = enqueue(b, b->upl, srcs, counts, stride_bytes, b->n_scans, nullptr, nullptr);
    if (rc == LOCGPU_OK)
        for (int s = 0; s < b->n_scans; ++s) b->counts[s] = (int)counts[s];
    return rc;
}

int upload_start_regions(locgpu_batch* b, BatchUploadState* st, float4* arena, int n_regions, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n,
                         const int* dst) {
    locgpu_ctx* ctx = b->ctx;
    if (!st || !dst || !arena || n < 0 || (n > 0 && (!srcs || !counts)) || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "pool upload: bad arguments");
    for (int s = 0; s < n; ++s) {
        if (dst[s] < 0 || dst[s] >= n_regions) return fail(ctx, LOCGPU_ERR_INVALID, "pool upload: bad region");
        if (counts[s] > (size_t)b->max_n) return fail(ctx, LOCGPU_ERR_INVALID, "pool upload: a scan has more points than the pool was created for");
        if (counts[s] && !srcs[s]) return fail(ctx, LOCGPU_ERR_INVALID, "pool upload: NULL scan pointer");
    }
    return enqueue(b, *st, srcs, counts, stride_bytes, n, dst, arena);
}

int upload_join_state(locgpu_ctx* ctx, BatchUploadState* st) {
    if (ctx->up && upload_host_busy(st)) {
        Uploader& u = *ctx->up;
        std::unique_lock<std::mutex> lk(u.m);
        u.cv_idle.wait(lk, [&] { return !upload_host_busy(st); });
    }
    if (st->rc != LOCGPU_OK) return fail(ctx, st->rc, st->err);
    return LOCGPU_OK;
}

void upload_drain(locgpu_ctx* ctx) {
    if (!ctx->up) return;
    Uploader& u = *ctx->up;
    std::unique_lock<std::mutex> lk(u.m);
    u.cv_idle.wait(lk, [&] { return u.queue.empty() && !u.busy; });
}

int upload_join_batch(locgpu_batch* b) { return upload_join_state(b->ctx, &b->upl); }

hipError_t upload_order_after(locgpu_batch* b, hipStream_t s) {
    if (!b->upl.done_valid) return hipSuccess;
    return hipStreamWaitEvent(s, b->upl.done, 0);
}

hipError_t upload_wait_landed(locgpu_batch* b) {
    if (!b->upl.done_valid) return hipSuccess;
    return hipEventSynchronize(b->upl.done);
}

void upload_free_batch(locgpu_batch* b) {
    locgpu_ctx* ctx = b->ctx;
    // a destroy while the service thread still packs scans for this batch (its own upload, or a pool job's into it): let it finish,
    // then let the copies land
    upload_drain(ctx);
    if (b->upl.done_valid && b->upl.done) (void)hipEventSynchronize(b->upl.done);
    if (b->upl.h_counts) (void)hipHostFree(b->upl.h_counts);
    if (b->upl.done) (void)hipEventDestroy(b->upl.done);
    b->upl.done = nullptr; b->upl.done_valid = false; b->upl.h_counts = nullptr; b->upl.rc = 0; b->upl.err.clear();
}

void upload_free_ctx(locgpu_ctx* ctx) {
    if (!ctx->up) return;
    Uploader& u = *ctx->up;
    if (u.started) {
        { std::lock_guard<std::mutex> lk(u.m); u.stop = true; }
        u.cv_work.notify_all();
        u.worker.join();
    }
    release_resources(u);
    delete ctx->up;
    ctx->up = nullptr;
}

}  // namespace locgpu
