// loc_lib_amd/csrc/batch_upload.hpp — host scans → HBM through pinned staging, off the caller's thread and off the compute streams.
//
// IcpRegistration::ScanMatch deep-copies its source cloud on every call (SetSource, icp_registration.cpp:221,252-265); for the
// batched mode that copy is 16 B/point over PCIe (472 MB for 256 full scans). The uploader packs the caller's strided points
// into small pinned slots with a few host threads and streams the slots to the batch's source array on a copy stream of its
// own, so the copy of batch i+1 runs under the Gauss–Newton loop of batch i (SDMA engine ‖ compute units).
//
// ONE uploader per context — copy stream, pinned slots, a service thread with a FIFO of requests (round 5: a request no longer waits
// for the one before it on the CALLER's thread — a scan pool's submit must return at once, its caller is what keeps the pool
// turning) — shared by all of the context's batches and pools; a batch (or a pool job) only owns the event behind its most recent
// upload and its status. Only as many slots as the upload has pieces are allocated (a one-scan batch pins 4 MB, not 64).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct locgpu_batch;
struct locgpu_ctx;

namespace locgpu {

struct BatchUploadState {  // per batch / per pool job
    hipEvent_t done = nullptr;  // recorded behind the last slot of the most recent upload
    bool done_valid = false;
    int* h_counts = nullptr;    // pinned copy of the per-scan point counts (batches only)
    int rc = 0;                 // status of the most recent upload once its host side is through (sticky until the next upload)
    std::string err;
    std::atomic<int> pending{0};  // requests queued or being packed: the service thread still reads the caller's clouds
};

struct UploadRequest {
    locgpu_batch* b = nullptr;
    BatchUploadState* st = nullptr;
    std::vector<const void*> srcs;  // the caller keeps the clouds alive until st->pending is back to 0
    std::vector<size_t> counts;
    std::vector<int> dst;           // scan i goes to region dst[i] of dst_base (empty: scan slot i of b, and the counts are copied too)
    float4* dst_base = nullptr;     // with dst: an arena of regions of b->max_n points (a scan pool's)
    size_t stride = 0;
    std::chrono::steady_clock::time_point t_start;  // diagnostics (LOCGPU_UPLOAD_DEBUG)
};

struct Uploader {  // per context
    static constexpr size_t kSlotPoints = 256 * 1024;  // 4 MB of float4 per pinned slot
    static constexpr int kSlotsPerThread = 2;
    hipStream_t stream = nullptr;  // copy stream
    int n_threads = 0;
    // pinned slots: touched by the service thread (and the packers it starts) only
    std::vector<float4*> h_slots;  // up to n_threads × kSlotsPerThread, allocated on demand
    std::vector<hipEvent_t> slot_ev;
    std::vector<char> slot_busy;   // the slot's event has been recorded: its last copy may still be reading it (kept across uploads)
    std::thread worker;
    std::mutex m;
    std::condition_variable cv_work, cv_idle;
    std::deque<UploadRequest> queue;
    bool busy = false, stop = false, started = false;
};

// n strided points (x, y, z as float32 at the start of each record) → float4 {x, y, z, 0}.
void pack_points(const char* base, size_t stride, size_t n, float4* dst);

// Queues packing + copying `srcs` into b's source array (and the counts into b's count array). Returns a locgpu_status.
int upload_start(locgpu_batch* b, const void* const* srcs, const size_t* counts, size_t stride_bytes);
// The same for `n` scans that go to the regions dst[0..n) of `arena` (regions of b->max_n points; a scan pool's source arena,
// scan_pool.hip): only the points move — the pool hands the counts to the device itself — and the event behind the last copy is
// st->done (one per pool job).
int upload_start_regions(locgpu_batch* b, BatchUploadState* st, float4* arena, int n_regions, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n,
                         const int* dst);
// Waits until the service thread has read every host cloud of the uploads that use `st` and enqueued their copies; returns the
// most recent one's status. The copies themselves may still be in flight: upload_order_after() makes a stream wait for them.
int upload_join_state(locgpu_ctx* ctx, BatchUploadState* st);
inline bool upload_host_busy(const BatchUploadState* st) { return st->pending.load(std::memory_order_acquire) > 0; }
// Waits until the context's uploader has nothing queued and nothing running.
void upload_drain(locgpu_ctx* ctx);
int upload_join_batch(locgpu_batch* b);
// Makes `s` wait for the most recent upload of b (no-op when there was none).
hipError_t upload_order_after(locgpu_batch* b, hipStream_t s);
// Blocks until the most recent upload of b has landed.
hipError_t upload_wait_landed(locgpu_batch* b);
void upload_free_batch(locgpu_batch* b);
void upload_free_ctx(locgpu_ctx* ctx);

}  // namespace locgpu
