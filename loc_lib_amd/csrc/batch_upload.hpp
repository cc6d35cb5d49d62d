// loc_lib_amd/csrc/batch_upload.hpp — host scans → HBM through pinned staging, off the caller's thread and off the compute streams.
//
// IcpRegistration::ScanMatch deep-copies its source cloud on every call (SetSource, icp_registration.cpp:221,252-265); for the
// batched mode that copy is 16 B/point over PCIe (472 MB for 256 full scans). The uploader packs the caller's strided points
// into small pinned slots with a few host threads and streams the slots to the batch's source array on a copy stream of its
// own, so the copy of batch i+1 runs under the Gauss–Newton loop of batch i (SDMA engine ‖ compute units).
//
// Round 3: ONE uploader per context — copy stream, pinned slots, worker — shared by all of its batches; a batch only
// owns the event behind its most recent upload and a pinned copy of its point counts. Only as many slots as the upload has
// pieces are allocated (a one-scan batch pins 4 MB, not 64), and a failed allocation leaves nothing half-built behind.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

struct locgpu_batch;
struct locgpu_ctx;

namespace locgpu {

struct Uploader {  // per context
    static constexpr size_t kSlotPoints = 256 * 1024;  // 4 MB of float4 per pinned slot
    static constexpr int kSlotsPerThread = 2;
    hipStream_t stream = nullptr;  // copy stream
    int n_threads = 0;
    std::vector<float4*> h_slots;  // up to n_threads × kSlotsPerThread pinned slots, allocated on demand
    std::vector<hipEvent_t> slot_ev;
    std::vector<char> slot_busy;   // the slot's event has been recorded: its last copy may still be reading it (kept across uploads)
    std::thread worker;
    bool worker_active = false;
    std::atomic<bool> worker_done{false};  // the worker has enqueued its last copy (it still has to be joined)
    locgpu_batch* current = nullptr;  // batch of the running upload
    struct BatchUploadState* current_st = nullptr;  // where its event and status live: the batch's own state, or a pool job's
    std::vector<int> dst;             // scan i of the upload goes to scan slot dst[i] of the batch (empty: slot i, and the counts are copied too)
    int rc = 0;
    std::string err;
    // arguments of the running upload (the caller keeps the clouds alive until upload_join)
    std::vector<const void*> srcs;
    std::vector<size_t> counts;
    size_t stride = 0;
    std::chrono::steady_clock::time_point t_start;  // diagnostics (LOCGPU_UPLOAD_DEBUG)
};

struct BatchUploadState {  // per batch
    hipEvent_t done = nullptr;  // recorded behind the last slot of the batch's most recent upload
    bool done_valid = false;
    int* h_counts = nullptr;    // pinned copy of the per-scan point counts
    int rc = 0;                 // status of the batch's most recent upload once joined (sticky until the next upload of the batch)
    std::string err;
};

// n strided points (x, y, z as float32 at the start of each record) → float4 {x, y, z, 0}.
void pack_points(const char* base, size_t stride, size_t n, float4* dst);

// Starts packing + copying `srcs` into b's source array. Returns a locgpu_status; on LOCGPU_OK the work continues on a worker thread.
int upload_start(locgpu_batch* b, const void* const* srcs, const size_t* counts, size_t stride_bytes);
// The same for `n` scans that go to the scan slots dst[0..n) of b (a scan pool's storage, scan_pool.hip): only the points move —
// the pool hands the counts to the device itself — and the event behind the last copy is st->done (one per pool job).
int upload_start_slots(locgpu_batch* b, BatchUploadState* st, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n, const int* dst);
// Is the upload whose state is `st` the one the context's worker was started for and not joined yet? *packing (optional): the
// worker is still reading the host clouds.
bool upload_running_for(locgpu_ctx* ctx, const BatchUploadState* st, bool* packing = nullptr);
// Waits until the context's worker has read every host cloud and enqueued every copy (of whatever batch it was working for);
// returns that upload's status. The copies themselves may still be in flight: upload_order_after() makes a stream wait for them.
int upload_join(locgpu_ctx* ctx);
// Same, but only when the running upload is b's (an upload of ANOTHER batch keeps running under b's alignment).
int upload_join_batch(locgpu_batch* b);
// Makes `s` wait for the most recent upload of b (no-op when there was none).
hipError_t upload_order_after(locgpu_batch* b, hipStream_t s);
// Blocks until the most recent upload of b has landed.
hipError_t upload_wait_landed(locgpu_batch* b);
void upload_free_batch(locgpu_batch* b);
void upload_free_ctx(locgpu_ctx* ctx);

}  // namespace locgpu
