// loc_lib_amd/csrc/batch_upload.hpp — host scans → HBM through pinned staging, off the caller's thread and off the compute stream.
//
// IcpRegistration::ScanMatch deep-copies its source cloud on every call (SetSource, icp_registration.cpp:221,252-265); for the
// batched mode that copy is 16 B/point over PCIe (472 MB for 256 full scans). The uploader packs the caller's strided points
// into small pinned slots with a few host threads and streams the slots to the batch's source array on a copy stream of its
// own, so the copy of batch i+1 runs under the Gauss–Newton loop of batch i (SDMA engine ‖ compute units).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <thread>
#include <vector>

struct locgpu_batch;

namespace locgpu {

struct BatchUploader {
    static constexpr size_t kSlotPoints = 256 * 1024;  // 4 MB of float4 per pinned slot
    static constexpr int kSlotsPerThread = 2;
    hipStream_t stream = nullptr;  // copy stream
    hipEvent_t done = nullptr;     // recorded behind the last slot of an upload
    bool done_valid = false;       // an upload has been enqueued since the batch was created
    int n_threads = 0;
    std::vector<float4*> h_slots;  // n_threads × kSlotsPerThread pinned slots
    std::vector<hipEvent_t> slot_ev;
    std::vector<char> slot_busy;   // the slot's event has been recorded: its last copy may still be reading it (kept across uploads)
    int* h_counts = nullptr;       // pinned copy of the per-scan point counts
    std::thread worker;
    bool worker_active = false;
    int rc = 0;
    std::string err;
    // arguments of the running upload (the caller keeps the clouds alive until upload_join)
    std::vector<const void*> srcs;
    std::vector<size_t> counts;
    size_t stride = 0;
};

// Starts packing + copying `srcs` into b's source array. Returns a locgpu_status; on LOCGPU_OK the work continues on a worker thread.
int upload_start(locgpu_batch* b, const void* const* srcs, const size_t* counts, size_t stride_bytes);
// Waits until the worker has read every host cloud and enqueued every copy; returns the upload's status. The copies themselves
// may still be in flight: upload_order_after() makes a stream wait for them.
int upload_join(locgpu_batch* b);
// Makes `s` wait for the most recent upload of b (no-op when there was none).
hipError_t upload_order_after(locgpu_batch* b, hipStream_t s);
void upload_free(locgpu_batch* b);

}  // namespace locgpu
