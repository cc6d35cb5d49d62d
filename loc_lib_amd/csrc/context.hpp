// loc_lib_amd/csrc/context.hpp — host-side state behind the opaque handles of include/locgpu.h.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/locgpu.h"
#include "batch_upload.hpp"
#include "device_math.hpp"
#include "grid_kernels.hpp"
#include "icp_kernels.hpp"

#include "host_worker.hpp"

struct NdtTable;  // ndt_kernels.hpp
namespace locgpu { struct IncNdtState; struct FilterScratch; }  // ndt_inc.hpp, cloud_filters.hpp

namespace locgpu { struct PendingTarget; }

struct locgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // = slot_stream[0]: target ingest, clouds, single-scan calls
    // Batches are dealt to kSlots compute streams in turn, so that an alignment begun on one batch (locgpu_*_align_batch_begin)
    // runs under the tail of the one begun before it: late Gauss–Newton iterations hold a handful of scans and leave most of the
    // chip idle. Three, because that is what pays (32 / 64 scans per batch: +14 % scans/s over two, nothing more with four) and
    // because three compute streams + the copy stream are the four hardware queues the runtime deals streams to in creation order.
    static constexpr int kSlots = 3;
    hipStream_t slot_stream[kSlots] = {};
    int next_slot = 0;
    hipStream_t copy_stream = nullptr;  // host → HBM copies of the batch uploader
    hipEvent_t foreign_ev = nullptr;    // ordering behind another context's stream when one of ITS clouds is an input here (cloud_input_ready)
    hipStream_t comm_stream = nullptr;  // every collective of the context, in host order (one communicator, one stream: no two at once)
    locgpu::Uploader* up = nullptr;     // host → HBM staging shared by the context's batches (batch_upload.hpp)
    locgpu::PendingTarget* target_scratch = nullptr;  // the previous ingest's host buffers, kept for the next one (locgpu_api.hip)
    locgpu::PendingTarget* pending_target = nullptr;  // locgpu_icp_set_target_cloud_async: a host tree build still running (locgpu_api.hip)
    std::string err;

    // ICP target: packed KD-tree in HBM (kdtree_build.cpp layout)
    uint2* d_tree = nullptr;
    size_t tree_slots = 0, num_leaves = 0, num_nodes = 0, num_points = 0;
    size_t tree_cap_slots = 0, leaf_cap = 0;  // capacities of d_tree / d_leaf_slots (grow-only)
    int depth = 0;
    bool tree_bounded = true;  // PackedKdTree::bounded: the fast search kernel may be used
    unsigned long long target_epoch = 0;  // bumped by every set_target: captured graphs of older targets are never replayed

    uint32_t* d_leaf_slots = nullptr;  // slot of every leaf, preorder (what the exact-search grid is built from)

    // exact-search grid over the tree's leaves (built on the device on first use of LOCGPU_SEARCH_GRID_EXACT)
    locgpu::GridView grid;
    locgpu::GridBuffers grid_buf;

    // BfnnRegistration target (bfnn.hip)
    float4* d_bfnn = nullptr;
    size_t bfnn_n = 0;

    // NDT target
    NdtTable* ndt = nullptr;
    locgpu::IncNdtState* inc = nullptr;  // incremental NDT voxel set (persists across set_target calls)
    locgpu_ndt_opts ndt_opts;

    locgpu::FilterScratch* filt = nullptr;  // workspaces of the cloud filters (cloud_filters.hip)
    void* loam = nullptr;                   // workspaces of the LOAM feature picker (loam_features.hip)

    // reusable one-scan batch for the single-scan entry points
    locgpu_batch* single = nullptr;
    size_t single_cap = 0;
    locgpu::HostWorker* worker = nullptr;  // locgpu_*_scan_match: host copy of the output cloud's fields beside the alignment

    // multi-GPU (locgpu_comm_init): RCCL communicator of the ranks that share sharded batches
    void* comm = nullptr;  // ncclComm_t
    int comm_rank = 0, comm_world = 1;

    // measurement
    int profile = 0;  // 0 off, 1 = events around search / fit+accumulate / solve, 2 = around the search stage only
    double prof_ms[3] = {0, 0, 0};
    long long prof_n[3] = {0, 0, 0};
    bool use_graph = false;  // replay a captured hipGraph of all GN iterations instead of eager chunks
    bool count_visits = false;
    unsigned long long* d_visits = nullptr;  // [4]: nodes, leaves, queries, distinct tree slots read (summed over launches)
    uint32_t* d_touched = nullptr;           // instrumented pass: one bit per tree slot
    size_t touched_words = 0;
    unsigned long long* d_search_stats = nullptr;  // [2]: queries searched / queries redone by the exact kernel
};

struct locgpu_batch {
    locgpu_ctx* ctx = nullptr;
    int slot = 0;
    hipStream_t stream = nullptr;  // ctx->slot_stream[slot]: everything the batch's alignments enqueue
    int n_scans = 0, max_n = 0, blocks_per_scan = 0;  // n_scans: the scans whose points THIS rank holds
    // Sharded batch (locgpu_batch_create_sharded): the batch has n_total scans, this rank holds the points of scans
    // [first, first + n_scans) — or, point-sharded, a slice of the points of every scan (first = 0, n_scans = n_total). Poses, flags
    // and normal equations exist for all n_total scans on every rank. Unsharded: n_total = n_scans, first = 0.
    int n_total = 0, first = 0;
    bool sharded = false;
    double* d_acc = nullptr;  // [n_total][kAccW] per-scan sums, all-reduced over the communicator (sharded batches only)
    size_t pitch = 0;  // n_scans * max_n
    float4* d_src = nullptr;
    int* d_counts = nullptr;
    locgpu::PoseState* d_state = nullptr;
    uint32_t* d_nn = nullptr;      // [5][pitch]
    const float4* d_src_ext = nullptr;           // one-scan batches: the points stay where the caller's cloud holds them (no copy into d_src); nullptr = d_src
    bool counters_clean = false;                 // the search stage's work-list counters are known to be zero (the last alignment ran to its end)
    int last_iterations = -1;                    // one-scan batches: iterations of the previous alignment run on this batch (-1: none yet) — sizes the next first chunk
    double* d_partials = nullptr;  // [n_scans][blocks_per_scan][kAccW]
    double* d_hb = nullptr;        // [n_scans][44]
    uint32_t* d_redo_list = nullptr;      // [pitch]
    unsigned int* d_redo_count = nullptr;  // [2]: the two lists' counters
    uint32_t* d_redo_list2 = nullptr;      // [pitch], allocated on first use of the grid search
    uint32_t* d_grid_qkey = nullptr;       // [pitch] grid search: tile of each query
    uint2* d_grid_sorted = nullptr;        // [pitch] grid search: {query, tile} in tile order
    uint32_t* d_grid_tile_count = nullptr; // [grid_tocc_cap + 1] grid search: queries per occupied tile of this batch's iteration
    void* d_grid_scan_temp = nullptr;      // [grid_scan_cap] workspace of the scan over them
    size_t grid_tocc_cap = 0, grid_scan_cap = 0;
    // hipGraph of {H2D state, max_iteration × (search, fit+accumulate, solve), D2H state}, keyed by the launch parameters
    hipGraphExec_t graph_exec = nullptr;       // {H2D state, first chunk of iterations, D2H state}
    hipGraphExec_t graph_exec_next = nullptr;  // {further chunk, D2H state}
    locgpu::GnParams graph_prm{};
    int graph_k = -1;
    float graph_alpha = 0.f;
    bool graph_ndt = false;
    const void* graph_target = nullptr;  // tree / NDT table the capture was made against
    unsigned long long graph_epoch = 0;
    float4* h_src = nullptr;               // pinned staging of the packed source (single-scan path only; reused across calls)
    hipEvent_t xyz_ev[8] = {};             // one-scan batch: the output cloud's pieces on their way back (write_output_cloud)
    locgpu::PoseState* h_state = nullptr;  // pinned
    // one-scan alignments paced from the host (locgpu_api.hip, align_finish): the solve kernel posts the state here after every
    // iteration — pinned COHERENT memory: [0] a finished scan's GnPostRecord, [kPostWord] call << 32 | iterations << 1 | done, [+1] checksum
    unsigned long long* h_post = nullptr;
    static constexpr int kPostWord = 16;  // in 8-byte words: behind the record, 16-byte aligned
    unsigned int post_call = 0;
    bool paced_tail = false;       // the last alignment was paced: up to pace_ahead idle launches may still be queued on `stream`
    hipEvent_t tail_ev = nullptr;  // ... behind which the next upload's copies are ordered (batch_upload.hip)
    double* h_hb = nullptr;                // pinned
    int* h_active = nullptr;               // pinned, [n_scans]: local indices of the scans still open at the last chunk boundary
    int* d_active = nullptr;               // its device copy (see SearchArgs::active)
    std::vector<int> counts;
    locgpu::BatchUploadState upl;          // event + pinned counts of locgpu_batch_upload_async (batch_upload.hpp)
    std::vector<hipEvent_t> events;        // profiling events of the batch's alignments (locgpu_profile_enable)
    hipEvent_t ev_ready = nullptr, ev_reduced = nullptr;  // sharded batches: compute stream ⇄ comm stream hand-over
    // an alignment begun with *_align_batch_begin and not yet finished
    struct Pending {
        bool active = false;
        locgpu::GnParams prm{};
        int k = 0;
        float alpha_eff = 0.f;
        bool ndt = false, graph = false;
        bool paced = false;  // one scan, eager: iterations are launched as the solve kernel posts its progress
        int launched = 0;
        size_t ev_used = 0;
        std::vector<double> init_poses;
    } pending;
};

namespace locgpu {

int fail(locgpu_ctx* ctx, int code, const std::string& msg);
bool hip_ok(locgpu_ctx* ctx, hipError_t e, const char* what);

// locgpu_api.hip
void ndt_free(locgpu_ctx* ctx);
// Device buffers + pinned result staging for n_scans scans of at most max_n points each; no points yet. n_total >= 0: a sharded batch.
int alloc_batch(locgpu_ctx* ctx, int n_scans, size_t max_n, locgpu_batch** out, int first = 0, int n_total = -1);
void free_batch(locgpu_batch* b);
// Validate the matcher's options against the context's target; fill the Gauss–Newton parameters of an alignment.
int check_icp(locgpu_ctx* ctx, const locgpu_icp_opts* o, GnParams& prm, int& k, float& alpha_eff);
int check_ndt(locgpu_ctx* ctx, GnParams& prm);
bool comm_all_reduce_f64(locgpu_ctx* ctx, double* buf, size_t count, hipStream_t s);

}  // namespace locgpu

#define LOCGPU_HIP(ctx, expr)                                                    \
    do {                                                                         \
        if (!locgpu::hip_ok((ctx), (expr), #expr)) return LOCGPU_ERR_NO_DEVICE;  \
    } while (0)
