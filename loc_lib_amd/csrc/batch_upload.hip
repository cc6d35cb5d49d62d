// loc_lib_amd/csrc/batch_upload.hip — see batch_upload.hpp.
#include "batch_upload.hpp"

#include <algorithm>
#include <atomic>
#include <cstring>

#include "context.hpp"

namespace locgpu {

namespace {

// strided points → float4 {x, y, z, 0}. Three float loads per point; the compiler vectorises the stride-12 and stride-16 cases.
void pack_points(const char* base, size_t stride, size_t n, float4* dst) {
    if (stride == sizeof(float4)) {  // already {x, y, z, w}: no kernel of the matcher reads w
        std::memcpy(dst, base, n * sizeof(float4));
        return;
    }
    for (size_t i = 0; i < n; ++i) {
        float v[3];
        std::memcpy(v, base + i * stride, 12);
        dst[i] = float4{v[0], v[1], v[2], 0.f};
    }
}

struct Unit { int scan; size_t off, len; };

bool ensure_resources(locgpu_batch* b) {
    BatchUploader& u = *b->up;
    locgpu_ctx* ctx = b->ctx;
    if (u.stream) return true;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    u.n_threads = (int)std::min<unsigned>(8u, std::max(2u, hw / 4));
    if (!hip_ok(ctx, hipStreamCreateWithFlags(&u.stream, hipStreamNonBlocking), "upload: hipStreamCreate") ||
        !hip_ok(ctx, hipEventCreateWithFlags(&u.done, hipEventDisableTiming), "upload: hipEventCreate") ||
        !hip_ok(ctx, hipHostMalloc((void**)&u.h_counts, (size_t)b->n_scans * sizeof(int)), "upload: hipHostMalloc counts"))
        return false;
    const int n_slots = u.n_threads * BatchUploader::kSlotsPerThread;
    u.h_slots.assign(n_slots, nullptr);
    u.slot_ev.assign(n_slots, nullptr);
    u.slot_busy.assign(n_slots, 0);
    for (int i = 0; i < n_slots; ++i)
        if (!hip_ok(ctx, hipHostMalloc((void**)&u.h_slots[i], BatchUploader::kSlotPoints * sizeof(float4)), "upload: hipHostMalloc slot") ||
            !hip_ok(ctx, hipEventCreateWithFlags(&u.slot_ev[i], hipEventDisableTiming), "upload: hipEventCreate"))
            return false;
    return true;
}

void run_upload(locgpu_batch* b) {
    BatchUploader& u = *b->up;
    (void)hipSetDevice(b->ctx->device);
    // work units: every scan in slot-sized pieces
    std::vector<Unit> units;
    for (int s = 0; s < b->n_scans; ++s)
        for (size_t o = 0; o < u.counts[s]; o += BatchUploader::kSlotPoints) units.push_back({s, o, std::min(BatchUploader::kSlotPoints, u.counts[s] - o)});
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto packer = [&](int t) {
        (void)hipSetDevice(b->ctx->device);
        int turn = 0;
        for (size_t i = next.fetch_add(1); i < units.size() && !failed.load(); i = next.fetch_add(1)) {
            const Unit& w = units[i];
            const int slot = t * BatchUploader::kSlotsPerThread + turn;
            // the slot's previous copy — of this upload or of the one before — has left it
            if (u.slot_busy[slot] && hipEventSynchronize(u.slot_ev[slot]) != hipSuccess) { failed = 1; break; }
            pack_points((const char*)u.srcs[w.scan] + w.off * u.stride, u.stride, w.len, u.h_slots[slot]);
            if (hipMemcpyAsync(b->d_src + (size_t)w.scan * b->max_n + w.off, u.h_slots[slot], w.len * sizeof(float4), hipMemcpyHostToDevice, u.stream) != hipSuccess ||
                hipEventRecord(u.slot_ev[slot], u.stream) != hipSuccess) { failed = 1; break; }
            u.slot_busy[slot] = 1;
            turn = (turn + 1) % BatchUploader::kSlotsPerThread;
        }
    };
    const int nt = (int)std::min<size_t>((size_t)u.n_threads, std::max<size_t>(1, units.size()));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(packer, t);
    packer(0);
    for (auto& t : th) t.join();
    bool ok = !failed.load();
    for (int s = 0; s < b->n_scans; ++s) u.h_counts[s] = (int)u.counts[s];
    ok = ok && hipMemcpyAsync(b->d_counts, u.h_counts, (size_t)b->n_scans * sizeof(int), hipMemcpyHostToDevice, u.stream) == hipSuccess &&
         hipEventRecord(u.done, u.stream) == hipSuccess;
    if (!ok) { u.rc = LOCGPU_ERR_NO_DEVICE; u.err = std::string("batch upload: ") + hipGetErrorString(hipGetLastError()); }
}

}  // namespace

int upload_start(locgpu_batch* b, const void* const* srcs, const size_t* counts, size_t stride_bytes) {
    locgpu_ctx* ctx = b->ctx;
    if (!srcs || !counts || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: bad arguments");
    for (int s = 0; s < b->n_scans; ++s) {
        if (counts[s] > (size_t)b->max_n) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: a scan has more points than the batch was created for");
        if (counts[s] && !srcs[s]) return fail(ctx, LOCGPU_ERR_INVALID, "batch upload: NULL scan pointer");
    }
    const int jrc = upload_join(b);  // one upload per batch at a time
    if (jrc != LOCGPU_OK) return jrc;
    if (!b->up) b->up = new BatchUploader();
    if (!ensure_resources(b)) return LOCGPU_ERR_OOM;
    BatchUploader& u = *b->up;
    // One upload of a batch at a time, end to end: the previous one's copies (same destination, same pinned counts) have landed.
    if (u.done_valid && !hip_ok(ctx, hipEventSynchronize(u.done), "batch upload: previous upload")) return LOCGPU_ERR_NO_DEVICE;
    // The previous contents of the source array may still be read by kernels enqueued on the compute stream (an align call
    // always synchronises before it returns, so in practice the stream is idle): order the copies behind them.
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
        (void)hipEventRecord(ev, ctx->stream);
        (void)hipStreamWaitEvent(u.stream, ev, 0);
        (void)hipEventDestroy(ev);
    }
    u.srcs.assign(srcs, srcs + b->n_scans);
    u.counts.assign(counts, counts + b->n_scans);
    u.stride = stride_bytes;
    u.rc = LOCGPU_OK;
    u.err.clear();
    for (int s = 0; s < b->n_scans; ++s) b->counts[s] = (int)counts[s];
    u.done_valid = true;
    u.worker_active = true;
    u.worker = std::thread(run_upload, b);
    return LOCGPU_OK;
}

int upload_join(locgpu_batch* b) {
    if (!b->up || !b->up->worker_active) return LOCGPU_OK;
    BatchUploader& u = *b->up;
    u.worker.join();
    u.worker_active = false;
    if (u.rc != LOCGPU_OK) return fail(b->ctx, u.rc, u.err);
    return LOCGPU_OK;
}

hipError_t upload_order_after(locgpu_batch* b, hipStream_t s) {
    if (!b->up || !b->up->done_valid) return hipSuccess;
    return hipStreamWaitEvent(s, b->up->done, 0);
}

void upload_free(locgpu_batch* b) {
    if (!b->up) return;
    BatchUploader& u = *b->up;
    if (u.worker_active) { u.worker.join(); u.worker_active = false; }
    if (u.stream) (void)hipStreamSynchronize(u.stream);
    for (float4* p : u.h_slots) if (p) (void)hipHostFree(p);
    for (hipEvent_t e : u.slot_ev) if (e) (void)hipEventDestroy(e);
    if (u.h_counts) (void)hipHostFree(u.h_counts);
    if (u.done) (void)hipEventDestroy(u.done);
    if (u.stream) (void)hipStreamDestroy(u.stream);
    delete b->up;
    b->up = nullptr;
}

}  // namespace locgpu
