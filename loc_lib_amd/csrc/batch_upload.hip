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
    u.stream = nullptr;  // the stream belongs to the context
}

// Copy stream + at least `want_slots` pinned slots (capped by the thread count). All or nothing: on a failure everything that
// was built is released again, so that the next call starts from scratch instead of finding a half-initialised uploader.
bool ensure_resources(locgpu_ctx* ctx, size_t want_slots) {
    Uploader& u = *ctx->up;
    if (!u.stream) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        u.n_threads = (int)std::min<unsigned>(8u, std::max(2u, hw / 4));
        u.stream = ctx->copy_stream;  // created with the context (see locgpu_create)
    }
    const size_t cap = (size_t)u.n_threads * Uploader::kSlotsPerThread;
    const size_t want = std::min(cap, std::max<size_t>(1, want_slots));
    while (u.h_slots.size() < want) {
        float4* p = nullptr;
        hipEvent_t ev = nullptr;
        if (!hip_ok(ctx, hipHostMalloc((void**)&p, Uploader::kSlotPoints * sizeof(float4)), "upload: hipHostMalloc slot") ||
            !hip_ok(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming), "upload: hipEventCreate")) {
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

void run_upload(locgpu_ctx* ctx) {
    Uploader& u = *ctx->up;
    locgpu_batch* b = u.current;
    const auto t_entry = std::chrono::steady_clock::now();
    (void)hipSetDevice(ctx->device);
    if (getenv("LOCGPU_UPLOAD_DEBUG"))
        fprintf(stderr, "[upload worker] started %.2f ms after upload_start, hipSetDevice took %.2f ms\n",
                std::chrono::duration<double, std::milli>(t_entry - u.t_start).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count());
    // work units: every scan in slot-sized pieces
    std::vector<Unit> units;
    const int n_up = (int)u.counts.size();
    const bool to_slots = !u.dst.empty();
    BatchUploadState& bst = *u.current_st;
    for (int s = 0; s < n_up; ++s)
        for (size_t o = 0; o < u.counts[s]; o += Uploader::kSlotPoints) units.push_back({s, o, std::min(Uploader::kSlotPoints, u.counts[s] - o)});
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    static const bool dbg = getenv("LOCGPU_UPLOAD_DEBUG") != nullptr;
    std::atomic<long long> wait_us{0}, pack_us{0};
    const auto t_begin = std::chrono::steady_clock::now();
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
            pack_points((const char*)u.srcs[w.scan] + w.off * u.stride, u.stride, w.len, u.h_slots[slot]);
            if (dbg) {
                wait_us += std::chrono::duration_cast<std::chrono::microseconds>(tb - ta).count();
                pack_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tb).count();
            }
            const size_t dst_scan = to_slots ? (size_t)u.dst[w.scan] : (size_t)w.scan;
            if (hipMemcpyAsync(b->d_src + dst_scan * b->max_n + w.off, u.h_slots[slot], w.len * sizeof(float4), hipMemcpyHostToDevice, u.stream) != hipSuccess ||
                hipEventRecord(u.slot_ev[slot], u.stream) != hipSuccess) { failed = 1; break; }
            u.slot_busy[slot] = 1;
            slot = slot + nt < n_slots ? slot + nt : t;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(packer, t);
    packer(0);
    for (auto& t : th) t.join();
    if (dbg)
        fprintf(stderr, "[upload worker] %d threads, %zu pieces: %.2f ms wall, per thread %.2f ms waiting for slots + %.2f ms packing\n", nt, units.size(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(), wait_us.load() / 1e3 / nt, pack_us.load() / 1e3 / nt);
    bool ok = !failed.load();
    if (!to_slots) {
        for (int s = 0; s < b->n_scans; ++s) bst.h_counts[s] = (int)u.counts[s];
        ok = ok && hipMemcpyAsync(b->d_counts, bst.h_counts, (size_t)b->n_scans * sizeof(int), hipMemcpyHostToDevice, u.stream) == hipSuccess;
    }
    ok = ok && hipEventRecord(bst.done, u.stream) == hipSuccess;
    if (!ok) { u.rc = LOCGPU_ERR_NO_DEVICE; u.err = std::string("batch upload: ") + hipGetErrorString(hipGetLastError()); }
    u.worker_done.store(true, std::memory_order_release);
}

}  // namespace

// Common tail of the two entry points: one upload per context at a time; `st` receives the event and the status.
static int start_worker(locgpu_batch* b, BatchUploadState& st, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n, const int* dst, size_t pieces) {
    locgpu_ctx* ctx = b->ctx;
    static const bool dbg = getenv("LOCGPU_UPLOAD_DEBUG") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // A failure of the upload that was running belongs to ITS batch / job (upload_join records it there: that batch's next align
    // or upload_wait reports it) and does not stop this one.
    (void)upload_join(ctx);
    const auto t1 = std::chrono::steady_clock::now();
    if (!ctx->up) ctx->up = new Uploader();
    if (!ensure_resources(ctx, pieces)) return LOCGPU_ERR_OOM;
    Uploader& u = *ctx->up;
    if (!st.done && !hip_ok(ctx, hipEventCreateWithFlags(&st.done, hipEventDisableTiming), "upload: hipEventCreate")) { st.done = nullptr; return LOCGPU_ERR_OOM; }
    if (!dst && !st.h_counts && !hip_ok(ctx, hipHostMalloc((void**)&st.h_counts, (size_t)b->n_scans * sizeof(int)), "upload: hipHostMalloc counts")) { st.h_counts = nullptr; return LOCGPU_ERR_OOM; }
    // One upload of a batch (or job) at a time, end to end: the previous one's copies (same destination, same pinned counts) have landed.
    if (st.done_valid && !hip_ok(ctx, hipEventSynchronize(st.done), "batch upload: previous upload")) return LOCGPU_ERR_NO_DEVICE;
    if (dbg) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[upload] join %.2f ms, previous upload of this batch landed %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    st.rc = LOCGPU_OK;
    st.err.clear();
    u.srcs.assign(srcs, srcs + n);
    u.counts.assign(counts, counts + n);
    u.dst.clear();
    if (dst) u.dst.assign(dst, dst + n);
    u.stride = stride_bytes;
    u.rc = LOCGPU_OK;
    u.err.clear();
    st.done_valid = true;
    u.current = b;
    u.current_st = &st;
    u.worker_active = true;
    u.worker_done.store(false, std::memory_order_relaxed);
    u.t_start = std::chrono::steady_clock::now();
    u.worker = std::thread(run_upload, ctx);
    return LOCGPU_OK;
}

int upload_start(locgpu_batch* b, const void* const* srcs, const size_t* counts, size_t stride_bytes) {
    locgpu_ctx* ctx = b->ctx;
    if (!srcs || !counts || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: bad arguments");
    size_t pieces = 0;
    for (int s = 0; s < b->n_scans; ++s) {
        if (counts[s] > (size_t)b->max_n) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: a scan has more points than the batch was created for");
        if (counts[s] && !srcs[s]) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: NULL scan pointer");
        pieces += (counts[s] + Uploader::kSlotPoints - 1) / Uploader::kSlotPoints;
    }
    // One alignment per batch at a time, and no upload into a batch whose alignment has been begun and not ended: the
    // chunks locgpu_align_batch_end enqueues later read d_src / d_counts / b->counts, which this upload would replace under them.
    // A caller with several alignments in flight rotates depth + 1 batches (bench.py) — the copy always goes to an idle one.
    // No ordering behind the batch's compute stream is needed beyond that: every ended alignment leaves its stream synchronised
    // (align_finish synchronises it on its error paths as well).
    if (b->pending.active) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: an alignment of this batch has been begun and not finished");
    const int rc = start_worker(b, b->upl, srcs, counts, stride_bytes, b->n_scans, nullptr, pieces);
    if (rc == LOCGPU_OK)
        for (int s = 0; s < b->n_scans; ++s) b->counts[s] = (int)counts[s];
    return rc;
}

int upload_start_slots(locgpu_batch* b, BatchUploadState* st, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n, const int* dst) {
    locgpu_ctx* ctx = b->ctx;
    if (!st || !dst || n < 0 || (n > 0 && (!srcs || !counts)) || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "pool upload: bad arguments");
    size_t pieces = 0;
    for (int s = 0; s < n; ++s) {
        if (dst[s] < 0 || dst[s] >= b->n_scans) return fail(ctx, LOCGPU_ERR_INVALID, "pool upload: bad slot");
        if (counts[s] > (size_t)b->max_n) return fail(ctx, LOCGPU_ERR_INVALID, "pool upload: a scan has more points than the pool was created for");
        if (counts[s] && !srcs[s]) return fail(ctx, LOCGPU_ERR_INVALID, "pool upload: NULL scan pointer");
        pieces += (counts[s] + Uploader::kSlotPoints - 1) / Uploader::kSlotPoints;
    }
    return start_worker(b, *st, srcs, counts, stride_bytes, n, dst, pieces);
}

bool upload_running_for(locgpu_ctx* ctx, const BatchUploadState* st, bool* packing) {
    const bool mine = ctx->up && ctx->up->worker_active && ctx->up->current_st == st;
    if (packing) *packing = mine && !ctx->up->worker_done.load(std::memory_order_acquire);
    return mine;
}

int upload_join(locgpu_ctx* ctx) {
    if (!ctx->up || !ctx->up->worker_active) return LOCGPU_OK;
    Uploader& u = *ctx->up;
    u.worker.join();
    u.worker_active = false;
    BatchUploadState* cur = u.current_st;
    u.current = nullptr;
    u.current_st = nullptr;
    if (u.rc != LOCGPU_OK) {
        // the failure stays with the batch it happened to: its source array is partly copied and its `done` event was not recorded
        // again, so every later use of that batch (align, upload_wait) must fail until a new upload replaces the scans
        if (cur) { cur->rc = u.rc; cur->err = u.err; cur->done_valid = false; }
        return fail(ctx, u.rc, u.err);
    }
    return LOCGPU_OK;
}

int upload_join_batch(locgpu_batch* b) {
    locgpu_ctx* ctx = b->ctx;
    if (upload_running_for(ctx, &b->upl)) return upload_join(ctx);
    if (b->upl.rc != LOCGPU_OK) return fail(ctx, b->upl.rc, b->upl.err);  // an earlier upload of this batch failed (joined on behalf of another call)
    return LOCGPU_OK;
}

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
    // a destroy while the worker still packs this batch's scans: let it finish, then let the copies land
    if (ctx->up && ctx->up->worker_active && ctx->up->current == b) (void)upload_join(ctx);  // the batch's own upload, or a pool job's into it
    if (b->upl.done_valid && b->upl.done) (void)hipEventSynchronize(b->upl.done);
    if (b->upl.h_counts) (void)hipHostFree(b->upl.h_counts);
    if (b->upl.done) (void)hipEventDestroy(b->upl.done);
    b->upl = BatchUploadState();
}

void upload_free_ctx(locgpu_ctx* ctx) {
    if (!ctx->up) return;
    Uploader& u = *ctx->up;
    if (u.worker_active) { u.worker.join(); u.worker_active = false; }
    release_resources(u);
    delete ctx->up;
    ctx->up = nullptr;
}

}  // namespace locgpu
