// loc_lib_amd/csrc/locgpu_api.hip — C ABI of include/locgpu.h: context, ICP target ingest, search, H/B, align, batch.
//
// Host control flow mirrors the reference's matcher (IcpRegistration, icp_registration.cpp): SetInputTarget
// builds the search structure once per map; ScanMatch runs the Gauss–Newton loop. Here the loop body is three
// kernel launches per iteration on one HIP stream and the convergence test lives on the device, so the host only
// reads back the small per-scan state every `kChunk` iterations.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "batch_upload.hpp"
#include "context.hpp"
#include "cloud_filters.hpp"
#include "kdtree_build.hpp"
#include "launch.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace locgpu;

// RCCL is bound at the first locgpu_comm_* call (dlopen), not at load time: a single-GPU process — the slam_demo front-end, the
// tests — never maps librccl and its dependencies (rocm_smi, roctx, rocprofiler-register). The entry points used:
namespace {
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string err;
};
Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // LOCGPU_RCCL_LIB names another library with the same six entry points: a site's own RCCL build, or the loopback double the
        // tests use to run two ranks as two threads on one GPU (tests/cpp/loopback_rccl.hip).
        const char* named = getenv("LOCGPU_RCCL_LIB");
        void* h = (named && *named) ? dlopen(named, RTLD_NOW | RTLD_LOCAL)
                                    : dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);  // a copy already mapped by the host process (e.g. PyTorch's) is reused
        if (!h && !(named && *named)) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { x.err = std::string("cannot load librccl: ") + dlerror(); return x; }
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(h, "ncclCommInitRank");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(h, "ncclCommDestroy");
        x.AllReduce = (decltype(x.AllReduce))dlsym(h, "ncclAllReduce");
        x.Broadcast = (decltype(x.Broadcast))dlsym(h, "ncclBroadcast");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(h, "ncclGetErrorString");
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.Broadcast && x.GetErrorString;
        if (!x.ok) x.err = "librccl lacks an expected entry point";
        return x;
    }();
    return r;
}
}  // namespace

// Sum of `count` doubles over the context's communicator, in place, on stream `s` (scan_pool.hip's exchange step).
bool locgpu::comm_all_reduce_f64(locgpu_ctx* ctx, double* buf, size_t count, hipStream_t s) {
    const ncclResult_t nr = rccl().AllReduce(buf, buf, count, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, s);
    if (nr == ncclSuccess) return true;
    fail(ctx, LOCGPU_ERR_NO_DEVICE, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr));
    return false;
}

namespace {
std::string g_create_err;
// GN iterations enqueued between two host reads of the convergence flags. Kernels of a finished scan return at once (device-side
// `done` flag), so running ahead costs ≈2 µs per empty launch while a host round trip costs tens of µs: the first chunk covers the
// typical alignment (7-8 iterations with the reference's eps), later ones are shorter.
constexpr int kFirstChunk = 8, kNextChunk = 4, kLongFirstChunk = 12;
// A one-scan alignment follows its first chunk with chunks of two: there a chunk boundary (read-back, host, relaunch ≈ 35 µs) costs
// about what two idle iterations do (3 dispatches of ≈4.6 µs each), and nine iterations — the common case beyond eight — then pay
// 35 + 15 µs instead of 35 + 45 (tools/single_scan_trace.py).
inline int next_chunk(const locgpu_batch* b) { return b->n_total == 1 ? 2 : kNextChunk; }
inline const float4* batch_src(const locgpu_batch* b) { return b->d_src_ext ? b->d_src_ext : b->d_src; }
// ... and sizes its FIRST chunk by the alignment it ran before: a front-end that matches every scan from a good prediction
// (Lio::AddCloud: 4-5 iterations per scan) used to pay three or four idle iterations, ≈15 µs each, in every call — a sixth of the
// match stage of the streaming loop (tools/stream_trace.py). One more than last time, between 3 and kFirstChunk; chunking never
// changes a result (the same kernels run on the same data in the same order), only where the host looks at the flags.
inline int first_chunk_len(const locgpu_batch* b) {
    if (b->n_total != 1 || b->sharded || b->last_iterations < 0) return kFirstChunk;
    // ... up to kLongFirstChunk when the last call needed more than eight: a chunk boundary there cost 33 µs + two idle iterations (round 5)
    return std::min(kLongFirstChunk, std::max(3, b->last_iterations + 1));
}
}  // namespace

namespace locgpu {
int fail(locgpu_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    else g_create_err = msg;
    return code;
}
bool hip_ok(locgpu_ctx* ctx, hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    fail(ctx, LOCGPU_ERR_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    return false;
}
}  // namespace locgpu

static void free_grid(locgpu_ctx* ctx) {
    locgpu::grid_free(ctx->grid_buf);
    ctx->grid = locgpu::GridView();
}

// Build the exact-search grid from the packed tree already in HBM (first use of LOCGPU_SEARCH_GRID_EXACT after a set_target).
static int ensure_grid(locgpu_ctx* ctx) {
    if (ctx->grid.pts) return LOCGPU_OK;
    std::string msg;
    const hipError_t e = locgpu::grid_build_device(ctx->d_tree, ctx->d_leaf_slots, ctx->num_leaves, ctx->stream, ctx->grid_buf, ctx->grid, msg);
    if (e != hipSuccess) {
        free_grid(ctx);
        if (!msg.empty()) return locgpu::fail(ctx, LOCGPU_ERR_INVALID, "grid search: " + msg);
        locgpu::hip_ok(ctx, e, "grid build");
        return LOCGPU_ERR_NO_DEVICE;
    }
    return LOCGPU_OK;
}

// Zero-fill of device memory that is COMPLETE when the call returns (see alloc_batch: a plain hipMemset is not, and is not ordered
// with the context's non-blocking streams either).
static bool fill_now(locgpu_ctx* ctx, void* p, size_t bytes, const char* what) {
    return hip_ok(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream), what) && hip_ok(ctx, hipStreamSynchronize(ctx->stream), what);
}

void locgpu::free_batch(locgpu_batch* b) {
    if (!b) return;
    upload_free_batch(b);
    for (hipEvent_t ev : b->events) (void)hipEventDestroy(ev);
    if (b->ev_ready) (void)hipEventDestroy(b->ev_ready);
    if (b->ev_reduced) (void)hipEventDestroy(b->ev_reduced);
    if (b->d_src) (void)hipFree(b->d_src);
    if (b->d_counts) (void)hipFree(b->d_counts);
    if (b->d_state) (void)hipFree(b->d_state);
    if (b->d_nn) (void)hipFree(b->d_nn);
    if (b->d_partials) (void)hipFree(b->d_partials);
    if (b->d_hb) (void)hipFree(b->d_hb);
    if (b->d_acc) (void)hipFree(b->d_acc);
    if (b->d_redo_list) (void)hipFree(b->d_redo_list);
    if (b->d_redo_count) (void)hipFree(b->d_redo_count);
    if (b->d_redo_list2) (void)hipFree(b->d_redo_list2);
    if (b->d_grid_qkey) (void)hipFree(b->d_grid_qkey);
    if (b->d_grid_sorted) (void)hipFree(b->d_grid_sorted);
    if (b->d_grid_tile_count) (void)hipFree(b->d_grid_tile_count);
    if (b->d_grid_scan_temp) (void)hipFree(b->d_grid_scan_temp);
    if (b->graph_exec) (void)hipGraphExecDestroy(b->graph_exec);
    if (b->graph_exec_next) (void)hipGraphExecDestroy(b->graph_exec_next);
    if (b->h_src) (void)hipHostFree(b->h_src);
    if (b->h_state) (void)hipHostFree(b->h_state);
    if (b->h_post) (void)hipHostFree(b->h_post);
    if (b->tail_ev) (void)hipEventDestroy(b->tail_ev);
    for (hipEvent_t e : b->xyz_ev)
        if (e) (void)hipEventDestroy(e);
    if (b->h_hb) (void)hipHostFree(b->h_hb);
    if (b->h_active) (void)hipHostFree(b->h_active);
    if (b->d_active) (void)hipFree(b->d_active);
    delete b;
}

static int target_join(locgpu_ctx* ctx, bool install = true);  // finishes a pending locgpu_icp_set_target_cloud_async (defined with it, below)
static void free_target_scratch(locgpu_ctx* ctx);  // the ingest buffers the context keeps between SetInputTarget calls (defined with PendingTarget, below)

extern "C" {

void locgpu_icp_opts_default(locgpu_icp_opts* o) {
    if (!o) return;
    o->method = LOCGPU_P2P;  // IcpOptions::method_{IcpMethod::P2P}, icp_registration.hpp:38
    o->max_iteration = 20;
    o->max_nn_distance = 1.0;
    o->max_plane_distance = 0.1;
    o->max_line_distance = 0.5;
    o->min_effective_pts = 10;
    o->eps = 1e-2;
    o->approximate = 1;
    o->ann_alpha = 0.1f;
    o->search_mode = LOCGPU_SEARCH_TREE_FAITHFUL;
}

void locgpu_ndt_opts_default(locgpu_ndt_opts* o) {
    if (!o) return;
    o->max_iteration = 20;
    o->voxel_size = 1.0;
    o->min_effective_pts = 10;
    o->min_pts_in_voxel = 3;
    o->eps = 1e-2;
    o->res_outlier_th = 20.0;
    o->nearby_type = 1;
    o->method = 1;
    o->capacity = 100000;
}

int locgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int locgpu_create(int device_id, locgpu_ctx** out) {
    if (!out) return fail(nullptr, LOCGPU_ERR_INVALID, "locgpu_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, LOCGPU_ERR_NO_DEVICE, std::string("no HIP device available (there is no CPU fallback): ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(nullptr, LOCGPU_ERR_INVALID, "locgpu_create: device_id out of range");
    auto* ctx = new locgpu_ctx();
    ctx->device = device_id;
    locgpu_ndt_opts_default(&ctx->ndt_opts);
    bool ok = hip_ok(nullptr, hipSetDevice(device_id), "hipSetDevice");
    for (int i = 0; ok && i < locgpu_ctx::kSlots; ++i) ok = hip_ok(nullptr, hipStreamCreateWithFlags(&ctx->slot_stream[i], hipStreamNonBlocking), "hipStreamCreate");
    // The copy stream and the communication stream are created HERE, right behind the compute streams: HIP deals streams to its
    // (four) hardware queues in creation order, and a copy stream that lands in the queue of a compute stream gets its H2D chunks
    // in between that stream's kernels only. (Seen with an RCCL communicator on the context: its internal streams shifted the
    // lazily created copy stream onto the first compute stream's queue — every upload that ran beside an alignment on that stream took
    // 31 ms instead of 14.)
    ok = ok && hip_ok(nullptr, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking), "hipStreamCreate");
    ok = ok && hip_ok(nullptr, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking), "hipStreamCreate");
    if (!ok) {
        for (hipStream_t st : ctx->slot_stream) if (st) (void)hipStreamDestroy(st);
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
        delete ctx;
        return LOCGPU_ERR_NO_DEVICE;
    }
    ctx->stream = ctx->slot_stream[0];
    if (!search_kernels_lds_ok()) {
        locgpu_destroy(ctx);
        return fail(nullptr, LOCGPU_ERR_INVALID, "locgpu_create: a search kernel of this build owns static LDS — its traversal stack would not start at LDS address 0 (search_walk.hpp)");
    }
    *out = ctx;
    return LOCGPU_OK;
}

void locgpu_destroy(locgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)target_join(ctx, false);
    free_target_scratch(ctx);
    for (hipStream_t st : ctx->slot_stream) if (st) (void)hipStreamSynchronize(st);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    delete ctx->worker;  // joins the helper thread (no job outlives the call that started it)
    free_batch(ctx->single);
    upload_free_ctx(ctx);
    if (ctx->comm) { (void)rccl().CommDestroy((ncclComm_t)ctx->comm); ctx->comm = nullptr; }
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->foreign_ev) (void)hipEventDestroy(ctx->foreign_ev);
    if (ctx->d_tree) (void)hipFree(ctx->d_tree);
    if (ctx->d_leaf_slots) (void)hipFree(ctx->d_leaf_slots);
    if (ctx->d_bfnn) (void)hipFree(ctx->d_bfnn);
    free_grid(ctx);
    if (ctx->d_visits) (void)hipFree(ctx->d_visits);
    if (ctx->d_touched) (void)hipFree(ctx->d_touched);
    if (ctx->d_search_stats) (void)hipFree(ctx->d_search_stats);
    ndt_free(ctx);
    filters_free(ctx);
    loam_free(ctx);
    for (hipStream_t st : ctx->slot_stream) if (st) (void)hipStreamDestroy(st);
    delete ctx;
}

const char* locgpu_last_error(const locgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// --------------------------------------------------------------------------------------------- target
// One target ingest's host side: the deep copy of the points, the packed tree, and — for the asynchronous entry points — the
// worker thread that builds it. The context keeps ONE of these between ingests (target_scratch): a keyframe front-end re-ingests
// its ≈35 k-point local map every few scans, and the vectors' capacity (≈1.4 MB: fresh mmaps and page faults per ingest) is
// worth keeping.
namespace locgpu {
struct PendingTarget {
    std::thread worker;
    std::vector<float> xyz;
    PackedKdTree tree;
    std::string err;
    bool ok = false;
};
}  // namespace locgpu

static locgpu::PendingTarget* take_target_scratch(locgpu_ctx* ctx) {
    locgpu::PendingTarget* p = ctx->target_scratch;
    ctx->target_scratch = nullptr;
    if (!p) p = new locgpu::PendingTarget();
    p->err.clear();
    p->ok = false;
    return p;
}

static void free_target_scratch(locgpu_ctx* ctx) {
    delete ctx->target_scratch;  // never holds a running worker (keep_target_scratch joins first)
    ctx->target_scratch = nullptr;
}

static void keep_target_scratch(locgpu_ctx* ctx, locgpu::PendingTarget* p) {
    if (p->worker.joinable()) p->worker.join();
    // a 10 M-point map's buffers (≈360 MB) go back to the allocator; a local map's stay
    if (ctx->target_scratch || p->xyz.capacity() > (size_t)3 << 21) delete p;
    else ctx->target_scratch = p;
}

// The reference's tree for `pts`, built on the host (kdtree_build.cpp).
static int build_host_tree(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, std::vector<float>& xyz, PackedKdTree& t) {
    if (!pts || n == 0 || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target: empty cloud or stride < 12");
    // deep copy (icp_registration.cpp:16 + kdtree.cpp:267 copy too): pack xyz
    xyz.resize(3 * n);
    const char* base = (const char*)pts;
    for (size_t i = 0; i < n; ++i) std::memcpy(&xyz[3 * i], base + i * stride_bytes, 12);
    std::string err;
    if (!build_packed_kdtree(xyz.data(), n, t, err)) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target: " + err);
    if (t.depth > 64) return fail(ctx, LOCGPU_ERR_DEPTH, "icp_set_target: KD-tree depth " + std::to_string(t.depth) + " exceeds the 64-entry traversal stack");
    return LOCGPU_OK;
}

// meta = {slots, leaves, nodes, points, depth, bounded}. The device buffers only ever grow: a streaming front-end re-ingests its
// local map every keyframe (lio.cpp:296-305) and must not pay a hipMalloc/hipFree pair (≈100 µs each) per ingest.
static int install_tree_meta(locgpu_ctx* ctx, const long long meta[6]) {
    for (hipStream_t st : ctx->slot_stream) LOCGPU_HIP(ctx, hipStreamSynchronize(st));  // nobody reads the old tree any more (a begun alignment must be finished first)
    free_grid(ctx);
    const size_t slots = (size_t)meta[0], leaves = (size_t)meta[1];
    if (slots + 2 > ctx->tree_cap_slots) {  // + the sentinel leaf behind the tree (search_walk.hpp)
        if (ctx->d_tree) { LOCGPU_HIP(ctx, hipFree(ctx->d_tree)); ctx->d_tree = nullptr; ctx->tree_cap_slots = 0; }
        const size_t cap = slots + slots / 4 + 1024;
        LOCGPU_HIP(ctx, hipMalloc((void**)&ctx->d_tree, cap * sizeof(uint64_t)));
        ctx->tree_cap_slots = cap;
    }
    if (leaves > ctx->leaf_cap) {
        if (ctx->d_leaf_slots) { LOCGPU_HIP(ctx, hipFree(ctx->d_leaf_slots)); ctx->d_leaf_slots = nullptr; ctx->leaf_cap = 0; }
        const size_t cap = leaves + leaves / 4 + 1024;
        LOCGPU_HIP(ctx, hipMalloc((void**)&ctx->d_leaf_slots, cap * sizeof(uint32_t)));
        ctx->leaf_cap = cap;
    }
    ctx->tree_slots = slots;
    ctx->num_leaves = leaves;
    ctx->num_nodes = (size_t)meta[2];
    ctx->num_points = (size_t)meta[3];
    ctx->depth = (int)meta[4];
    ctx->tree_bounded = meta[5] != 0;
    ctx->target_epoch++;
    return LOCGPU_OK;
}

// The sentinel leaf behind the packed tree (two slots at index tree_slots): what a lane of the search kernel "visits" when it has
// no node to visit. Its coordinates are so large that the squared distance overflows to +inf for every sane query.
static hipError_t write_sentinel_leaf(locgpu_ctx* ctx) {
    static const uint32_t leaf[4] = {0x7F61B1E6u /* 3.0e38f */, 0xC0000000u, 0x7F61B1E6u, 0x7F61B1E6u};
    return hipMemcpyAsync(ctx->d_tree + ctx->tree_slots, leaf, sizeof(leaf), hipMemcpyHostToDevice, ctx->stream);
}

// ---- SetInputTarget with the host build off the caller's thread (locgpu_icp_set_target_cloud_async) ----
// The mean-split tree is built on the host (its float32 sums are sequential by definition), 0.5–0.7 ms for a 35 k-pt local map. A
// streaming front-end that re-ingests its local map every keyframe (lio.cpp:296-305) has work to do in the meantime — upload and
// filter the next scan — so the build may run on a worker thread. Only the BUILD does: the worker touches its own copy of the points,
// its own PackedKdTree and the process-wide build pool, nothing of HIP and nothing of the context; every HIP call of the ingest
// (buffers, H2D, sentinel) is made by the caller's thread in target_join(), which every entry point that reads the target calls first.

static int install_built_tree(locgpu_ctx* ctx, const PackedKdTree& t) {
    const long long meta[6] = {(long long)t.slots.size(), (long long)t.num_leaves, (long long)t.num_nodes, (long long)t.num_points, t.depth, t.bounded ? 1 : 0};
    const int rc = install_tree_meta(ctx, meta);
    if (rc != LOCGPU_OK) return rc;
    LOCGPU_HIP(ctx, hipMemcpyAsync(ctx->d_tree, t.slots.data(), t.slots.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    LOCGPU_HIP(ctx, hipMemcpyAsync(ctx->d_leaf_slots, t.leaf_slots.data(), t.leaf_slots.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    LOCGPU_HIP(ctx, write_sentinel_leaf(ctx));
    LOCGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LOCGPU_OK;
}

// Finishes a pending asynchronous ingest (no-op without one). install = false: only wait for the worker (context teardown, or a new
// target that supersedes the pending one).
static int target_join(locgpu_ctx* ctx, bool install) {
    if (!ctx || !ctx->pending_target) return LOCGPU_OK;
    locgpu::PendingTarget* p = ctx->pending_target;
    ctx->pending_target = nullptr;
    if (p->worker.joinable()) p->worker.join();
    int rc = LOCGPU_OK;
    if (install) {
        if (!p->ok) rc = fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target: " + p->err);
        else if (p->tree.depth > 64) rc = fail(ctx, LOCGPU_ERR_DEPTH, "icp_set_target: KD-tree depth " + std::to_string(p->tree.depth) + " exceeds the 64-entry traversal stack");
        else if (hipSetDevice(ctx->device) != hipSuccess) rc = LOCGPU_ERR_NO_DEVICE;
        else rc = install_built_tree(ctx, p->tree);
    }
    keep_target_scratch(ctx, p);
    return rc;
}

int locgpu_icp_set_target(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    (void)target_join(ctx, false);  // a pending asynchronous ingest is superseded
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    static const bool times = getenv("LOCGPU_INGEST_TIMES") != nullptr;  // diagnostic: phase times on stderr
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!times) return;
        const auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[locgpu ingest] %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    locgpu::PendingTarget* p = take_target_scratch(ctx);
    int rc = build_host_tree(ctx, pts, n, stride_bytes, p->xyz, p->tree);
    lap("host build");
    if (rc == LOCGPU_OK) {
        rc = install_built_tree(ctx, p->tree);
        lap("device buffers + H2D");
    }
    keep_target_scratch(ctx, p);
    return rc;
}

// The same with the host build on a worker thread (see PendingTarget): returns once the points have been copied.
int locgpu_icp_set_target_async(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!pts || n == 0 || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target: empty cloud or stride < 12");
    (void)target_join(ctx, false);  // an earlier pending ingest is superseded
    locgpu::PendingTarget* p = take_target_scratch(ctx);
    p->xyz.resize(3 * n);  // the deep copy of SetInputTarget (icp_registration.cpp:16)
    const char* base = (const char*)pts;
    for (size_t i = 0; i < n; ++i) std::memcpy(&p->xyz[3 * i], base + i * stride_bytes, 12);
    p->worker = std::thread([p, n] { p->ok = build_packed_kdtree(p->xyz.data(), n, p->tree, p->err); });
    ctx->pending_target = p;
    return LOCGPU_OK;
}

// Collective over the context's communicator: rank `root` builds the tree from its `pts` (the other ranks' pts/n are ignored)
// and broadcasts the packed tree over xGMI — one host build per node instead of one per GPU.
int locgpu_icp_set_target_bcast(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, int root) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!ctx->comm) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target_bcast: locgpu_comm_init has not been called");
    if (root < 0 || root >= ctx->comm_world) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target_bcast: bad root");
    (void)target_join(ctx, false);  // a pending asynchronous ingest is superseded
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    hipStream_t s = ctx->stream;
    PackedKdTree t;
    std::vector<float> xyz;
    long long meta[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // [6] = the root's status
    int rc = LOCGPU_OK;
    if (ctx->comm_rank == root) {
        rc = build_host_tree(ctx, pts, n, stride_bytes, xyz, t);
        if (rc == LOCGPU_OK) { meta[0] = (long long)t.slots.size(); meta[1] = (long long)t.num_leaves; meta[2] = (long long)t.num_nodes; meta[3] = (long long)t.num_points; meta[4] = t.depth; meta[5] = t.bounded ? 1 : 0; }
        meta[6] = rc;
    }
    long long* d_meta = nullptr;
    LOCGPU_HIP(ctx, hipMalloc((void**)&d_meta, sizeof(meta)));
    bool ok = hip_ok(ctx, hipMemcpyAsync(d_meta, meta, sizeof(meta), hipMemcpyHostToDevice, s), "bcast meta H2D");
    ok = ok && rccl().Broadcast(d_meta, d_meta, sizeof(meta), ncclChar, root, comm, s) == ncclSuccess;
    ok = ok && hip_ok(ctx, hipMemcpyAsync(meta, d_meta, sizeof(meta), hipMemcpyDeviceToHost, s), "bcast meta D2H") && hip_ok(ctx, hipStreamSynchronize(s), "sync");
    (void)hipFree(d_meta);
    if (!ok) return fail(ctx, LOCGPU_ERR_NO_DEVICE, "icp_set_target_bcast: broadcast of the tree header failed");
    if (meta[6] != LOCGPU_OK) return ctx->comm_rank == root ? (int)meta[6] : fail(ctx, (int)meta[6], "icp_set_target_bcast: the root rank could not build the tree");
    rc = install_tree_meta(ctx, meta);
    {
        // collective error exit: a rank that could not make room for the tree must not leave the others waiting in the broadcast
        int* d_rc = nullptr;
        int all_rc = rc;
        bool okc = hip_ok(ctx, hipMalloc((void**)&d_rc, sizeof(int)), "bcast status") &&
                   hip_ok(ctx, hipMemcpyAsync(d_rc, &rc, sizeof(int), hipMemcpyHostToDevice, s), "bcast status H2D");
        okc = okc && rccl().AllReduce(d_rc, d_rc, 1, ncclInt, ncclMin, comm, s) == ncclSuccess;  // status codes are <= 0
        okc = okc && hip_ok(ctx, hipMemcpyAsync(&all_rc, d_rc, sizeof(int), hipMemcpyDeviceToHost, s), "bcast status D2H") && hip_ok(ctx, hipStreamSynchronize(s), "sync");
        if (d_rc) (void)hipFree(d_rc);
        if (!okc) return fail(ctx, LOCGPU_ERR_NO_DEVICE, "icp_set_target_bcast: status exchange failed");
        if (rc != LOCGPU_OK) return rc;
        if (all_rc != LOCGPU_OK) return fail(ctx, all_rc, "icp_set_target_bcast: another rank could not allocate the tree buffers");
    }
    if (ctx->comm_rank == root) {
        LOCGPU_HIP(ctx, hipMemcpyAsync(ctx->d_tree, t.slots.data(), t.slots.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        LOCGPU_HIP(ctx, hipMemcpyAsync(ctx->d_leaf_slots, t.leaf_slots.data(), t.leaf_slots.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    }
    if (rccl().Broadcast(ctx->d_tree, ctx->d_tree, ctx->tree_slots * sizeof(uint64_t), ncclChar, root, comm, s) != ncclSuccess ||
        rccl().Broadcast(ctx->d_leaf_slots, ctx->d_leaf_slots, ctx->num_leaves * sizeof(uint32_t), ncclChar, root, comm, s) != ncclSuccess)
        return fail(ctx, LOCGPU_ERR_NO_DEVICE, "icp_set_target_bcast: broadcast of the tree failed");
    LOCGPU_HIP(ctx, write_sentinel_leaf(ctx));
    LOCGPU_HIP(ctx, hipStreamSynchronize(s));
    return LOCGPU_OK;
}

int locgpu_icp_target_info(const locgpu_ctx* ctx, int64_t out[4]) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    { const int jrc = target_join(const_cast<locgpu_ctx*>(ctx)); if (jrc != LOCGPU_OK) return jrc; }
    out[0] = (int64_t)ctx->num_leaves;
    out[1] = (int64_t)ctx->num_nodes;
    out[2] = ctx->depth;
    out[3] = (int64_t)(ctx->tree_slots * sizeof(uint64_t));
    return ctx->d_tree ? LOCGPU_OK : LOCGPU_ERR_NO_TARGET;
}

// --------------------------------------------------------------------------------------------- k-NN
int locgpu_knn(locgpu_ctx* ctx, const float* queries, size_t nq, int k, int approximate, float alpha, int search_mode, int32_t* out_idx,
               uint32_t* visits) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    { const int jrc = target_join(ctx); if (jrc != LOCGPU_OK) return jrc; }
    if (!ctx->d_tree) return fail(ctx, LOCGPU_ERR_NO_TARGET, "knn: no target set");
    if (!queries || !out_idx || k < 1 || k > 8) return fail(ctx, LOCGPU_ERR_INVALID, "knn: bad arguments (1 <= k <= 8)");
    if (search_mode != LOCGPU_SEARCH_TREE_FAITHFUL && search_mode != LOCGPU_SEARCH_GRID_EXACT) return fail(ctx, LOCGPU_ERR_INVALID, "knn: unknown search mode");
    if ((size_t)k > ctx->num_leaves) return fail(ctx, LOCGPU_ERR_K_TOO_LARGE, "knn: k larger than the number of tree leaves");
    if (nq == 0) return LOCGPU_OK;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    const bool grid = search_mode == LOCGPU_SEARCH_GRID_EXACT;
    if (grid) {
        if (k != 1 && k != 5) return fail(ctx, LOCGPU_ERR_INVALID, "knn: grid search supports k = 1 or 5");
        const int grc = ensure_grid(ctx);
        if (grc != LOCGPU_OK) return grc;
    }
    float* d_q = nullptr;
    int32_t* d_out = nullptr;
    uint32_t* d_vis = nullptr;
    unsigned int* d_flag = nullptr;
    int rc = LOCGPU_OK;
    auto cleanup = [&]() { if (d_q) (void)hipFree(d_q); if (d_out) (void)hipFree(d_out); if (d_vis) (void)hipFree(d_vis); if (d_flag) (void)hipFree(d_flag); };
    if (!hip_ok(ctx, hipMalloc((void**)&d_q, nq * 12), "hipMalloc") || !hip_ok(ctx, hipMalloc((void**)&d_out, nq * k * 4), "hipMalloc") ||
        !hip_ok(ctx, hipMalloc((void**)&d_flag, 4), "hipMalloc") || (visits && !hip_ok(ctx, hipMalloc((void**)&d_vis, nq * 8), "hipMalloc"))) {
        cleanup();
        return LOCGPU_ERR_OOM;
    }
    hipStream_t s = ctx->stream;
    if (!hip_ok(ctx, hipMemcpyAsync(d_q, queries, nq * 12, hipMemcpyHostToDevice, s), "H2D")) rc = LOCGPU_ERR_NO_DEVICE;
    if (rc == LOCGPU_OK && grid) {
        // grid kernel first; the queries it cannot settle (out_idx[i*k] == -2) are answered by the exact tree traversal
        unsigned int flagged = 0;
        (void)hipMemsetAsync(d_flag, 0, 4, s);
        if (!launch_knn_grid_query(ctx->grid, ctx->d_tree, d_q, nq, k, d_out, d_flag, s)) rc = fail(ctx, LOCGPU_ERR_INVALID, "knn: unsupported k");
        if (rc == LOCGPU_OK && (!hip_ok(ctx, hipMemcpyAsync(out_idx, d_out, nq * k * 4, hipMemcpyDeviceToHost, s), "D2H") ||
                                !hip_ok(ctx, hipMemcpyAsync(&flagged, d_flag, 4, hipMemcpyDeviceToHost, s), "D2H") ||
                                !hip_ok(ctx, hipStreamSynchronize(s), "sync")))
            rc = LOCGPU_ERR_NO_DEVICE;
        if (rc == LOCGPU_OK && visits) std::memset(visits, 0, nq * 8);
        if (rc == LOCGPU_OK && flagged) {
            std::vector<int32_t> tmp(nq * k);
            if (!launch_knn_query(ctx->d_tree, ctx->depth, d_q, nq, k, 1.0f, d_out, nullptr, s)) rc = fail(ctx, LOCGPU_ERR_DEPTH, "knn: unsupported k/depth");
            if (rc == LOCGPU_OK && (!hip_ok(ctx, hipMemcpyAsync(tmp.data(), d_out, nq * k * 4, hipMemcpyDeviceToHost, s), "D2H") ||
                                    !hip_ok(ctx, hipStreamSynchronize(s), "sync")))
                rc = LOCGPU_ERR_NO_DEVICE;
            if (rc == LOCGPU_OK)
                for (size_t i = 0; i < nq; ++i)
                    if (out_idx[i * k] == -2)
                        for (int j = 0; j < k; ++j) out_idx[i * k + j] = tmp[i * k + j];
        }
        cleanup();
        return rc;
    }
    if (rc == LOCGPU_OK && !launch_knn_query(ctx->d_tree, ctx->depth, d_q, nq, k, approximate ? alpha : 1.0f, d_out, d_vis, s))
        rc = fail(ctx, LOCGPU_ERR_DEPTH, "knn: unsupported k/depth");
    if (rc == LOCGPU_OK && !hip_ok(ctx, hipGetLastError(), "knn launch")) rc = LOCGPU_ERR_NO_DEVICE;
    if (rc == LOCGPU_OK && !hip_ok(ctx, hipMemcpyAsync(out_idx, d_out, nq * k * 4, hipMemcpyDeviceToHost, s), "D2H")) rc = LOCGPU_ERR_NO_DEVICE;
    if (rc == LOCGPU_OK && visits && !hip_ok(ctx, hipMemcpyAsync(visits, d_vis, nq * 8, hipMemcpyDeviceToHost, s), "D2H")) rc = LOCGPU_ERR_NO_DEVICE;
    if (!hip_ok(ctx, hipStreamSynchronize(s), "sync") && rc == LOCGPU_OK) rc = LOCGPU_ERR_NO_DEVICE;
    cleanup();
    return rc;
}

// --------------------------------------------------------------------------------------------- batches
}  // extern "C"

// Device buffers + pinned result staging for n_scans scans of at most max_n points each; no points yet.
int locgpu::alloc_batch(locgpu_ctx* ctx, int n_scans, size_t max_n, locgpu_batch** out, int first, int n_total) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    *out = nullptr;
    const bool sharded = n_total >= 0;
    if (!sharded) n_total = n_scans;
    // a sharded batch may hold NO scan on this rank (more ranks than scans): it then only contributes zeros to the exchange
    if (n_scans < 0 || (n_scans == 0 && !sharded) || first < 0 || first + n_scans > n_total) return fail(ctx, LOCGPU_ERR_INVALID, "batch_create: bad arguments");
    if (n_total > 65535) return fail(ctx, LOCGPU_ERR_INVALID, "batch_create: at most 65535 scans per batch");
    if (n_scans > 65535) return fail(ctx, LOCGPU_ERR_INVALID, "batch_create: at most 65535 scans per batch");
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    if (n_scans == 0) max_n = 1;
    if (max_n == 0 || max_n > 0x7FFFFF00u || (size_t)n_scans * max_n > 0xFFFFFFF0ull)
        return fail(ctx, LOCGPU_ERR_INVALID, "batch_create: scans are empty or the batch exceeds 2^32 points");
    auto* b = new locgpu_batch();
    b->ctx = ctx;
    b->slot = ctx->next_slot;
    ctx->next_slot = (ctx->next_slot + 1) % locgpu_ctx::kSlots;
    b->stream = ctx->slot_stream[b->slot];
    b->n_scans = n_scans;
    b->n_total = n_total;
    b->first = first;
    b->sharded = sharded;
    b->max_n = (int)max_n;
    b->blocks_per_scan = (int)((max_n + kBlock - 1) / kBlock);
    b->pitch = (size_t)n_scans * max_n;
    b->counts.assign(n_scans, 0);
    bool ok = hip_ok(ctx, hipMalloc((void**)&b->d_src, std::max<size_t>(b->pitch, 1) * sizeof(float4)), "hipMalloc src") &&
              hip_ok(ctx, hipMalloc((void**)&b->d_counts, std::max(n_scans, 1) * sizeof(int)), "hipMalloc counts") &&
              hip_ok(ctx, hipMalloc((void**)&b->d_state, n_total * sizeof(PoseState)), "hipMalloc state") &&
              (!sharded || hip_ok(ctx, hipMalloc((void**)&b->d_acc, (size_t)kFirstChunk * n_total * kAccW * sizeof(double)), "hipMalloc acc")) &&  // one slot per iteration of a chunk
              hip_ok(ctx, hipMalloc((void**)&b->d_nn, 5 * std::max<size_t>(b->pitch, 1) * sizeof(uint32_t)), "hipMalloc nn") &&
              hip_ok(ctx, hipMalloc((void**)&b->d_partials, (size_t)std::max(n_scans, 1) * b->blocks_per_scan * kAccW * sizeof(double)), "hipMalloc partials") &&
              hip_ok(ctx, hipMalloc((void**)&b->d_hb, (size_t)n_total * 44 * sizeof(double)), "hipMalloc hb") &&
              hip_ok(ctx, hipMalloc((void**)&b->d_redo_list, (getenv("LOCGPU_STAMP") ? 2 : 1) * std::max<size_t>(b->pitch, 1) * sizeof(uint32_t)), "hipMalloc redo") &&  // diagnostic build: + per-query trip counts
              hip_ok(ctx, hipMalloc((void**)&b->d_redo_list2, std::max<size_t>(b->pitch, 1) * sizeof(uint32_t)), "hipMalloc redo2") &&  // deep pass / grid search: second work list
              hip_ok(ctx, hipMalloc((void**)&b->d_redo_count, 4 * sizeof(unsigned int)), "hipMalloc redo") &&  // [0] redo list, [1] deep list, [2..3] spare
              hip_ok(ctx, hipHostMalloc((void**)&b->h_state, n_total * sizeof(PoseState)), "hipHostMalloc state") &&
              hip_ok(ctx, hipHostMalloc((void**)&b->h_hb, (size_t)n_total * 44 * sizeof(double)), "hipHostMalloc hb") &&
              hip_ok(ctx, hipHostMalloc((void**)&b->h_active, (size_t)std::max(n_scans, 1) * sizeof(int)), "hipHostMalloc active") &&
              hip_ok(ctx, hipMalloc((void**)&b->d_active, (size_t)std::max(n_scans, 1) * sizeof(int)), "hipMalloc active") &&
              // hipMemset runs on the NULL stream and returns before it has run; the context's streams are non-blocking, i.e. NOT ordered
              // behind it — the first upload of the scan counts could be overtaken by this very fill (the first alignment of a fresh
              // batch then saw a scan of zero points and ran its 20 iterations on nothing; once in ≈1000 first calls, found by
              // tools/fuzz_align.py --cases 120 in round 4). fill_now: on the context's stream, and waited for.
              fill_now(ctx, b->d_counts, std::max(n_scans, 1) * sizeof(int), "hipMemset counts") &&
              fill_now(ctx, b->d_redo_count, 4 * sizeof(unsigned int), "hipMemset redo");  // kept zero between searches by gn_solve_kernel
    if (!ok) { free_batch(b); return LOCGPU_ERR_OOM; }
    *out = b;
    return LOCGPU_OK;
}

extern "C" {

// Batch with the given scans resident when the call returns (deep copy of the sources, icp_registration.cpp:259).
static int make_batch(locgpu_ctx* ctx, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n_scans, locgpu_batch** out,
                      int first = 0, int n_total = -1) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    *out = nullptr;
    if (n_scans < 0 || (n_scans > 0 && (!srcs || !counts)) || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "batch_create: bad arguments");
    size_t max_n = 0;
    for (int s = 0; s < n_scans; ++s) max_n = std::max(max_n, counts[s]);
    for (int s = 0; s < n_scans; ++s)
        if (!srcs[s] && counts[s]) return fail(ctx, LOCGPU_ERR_INVALID, "batch_create: NULL scan pointer");
    locgpu_batch* b = nullptr;
    int rc = alloc_batch(ctx, n_scans, max_n, &b, first, n_total);
    if (rc != LOCGPU_OK) return rc;
    if (n_scans > 0) {
        rc = upload_start(b, srcs, counts, stride_bytes);
        if (rc == LOCGPU_OK) rc = upload_join_batch(b);
        if (rc == LOCGPU_OK && !hip_ok(ctx, upload_wait_landed(b), "batch_create: H2D")) rc = LOCGPU_ERR_NO_DEVICE;
    }
    if (rc != LOCGPU_OK) { free_batch(b); return rc; }
    *out = b;
    return LOCGPU_OK;
}

int locgpu_batch_create(locgpu_ctx* ctx, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n_scans, locgpu_batch** out) {
    return make_batch(ctx, srcs, counts, stride_bytes, n_scans, out);
}

int locgpu_batch_create_empty(locgpu_ctx* ctx, int n_scans, size_t max_points_per_scan, locgpu_batch** out) {
    return alloc_batch(ctx, n_scans, max_points_per_scan, out);
}

int locgpu_batch_upload_async(locgpu_batch* b, const void* const* srcs, const size_t* counts, size_t stride_bytes) {
    if (!b) return LOCGPU_ERR_INVALID;
    LOCGPU_HIP(b->ctx, hipSetDevice(b->ctx->device));
    return upload_start(b, srcs, counts, stride_bytes);
}

int locgpu_batch_upload_wait(locgpu_batch* b) {
    if (!b) return LOCGPU_ERR_INVALID;
    return upload_join_batch(b);
}

int locgpu_batch_create_sharded(locgpu_ctx* ctx, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n_local, int first_scan,
                                int n_total, locgpu_batch** out) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (n_total <= 0) return fail(ctx, LOCGPU_ERR_INVALID, "batch_create_sharded: n_total must be positive");
    // Without a communicator nobody else can contribute the scans this rank does not hold: their sums would silently stay zero.
    if (!ctx->comm && !(first_scan == 0 && n_local == n_total))
        return fail(ctx, LOCGPU_ERR_INVALID, "batch_create_sharded: this rank holds only part of the batch and locgpu_comm_init has not been called");
    const int rc = make_batch(ctx, srcs, counts, stride_bytes, n_local, out, first_scan, n_total);
    if (rc != LOCGPU_OK) return rc;
    locgpu_batch* b = *out;
    if (!hip_ok(ctx, hipEventCreateWithFlags(&b->ev_ready, hipEventDisableTiming), "batch_create_sharded: hipEventCreate") ||
        !hip_ok(ctx, hipEventCreateWithFlags(&b->ev_reduced, hipEventDisableTiming), "batch_create_sharded: hipEventCreate")) {
        free_batch(b);
        *out = nullptr;
        return LOCGPU_ERR_OOM;
    }
    return LOCGPU_OK;
}

int locgpu_comm_unique_id(void* id_out) {
    if (!id_out) return LOCGPU_ERR_INVALID;
    static_assert(LOCGPU_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "locgpu.h and rccl.h disagree on the id size");
    ncclUniqueId id;
    if (!rccl().ok || rccl().GetUniqueId(&id) != ncclSuccess) return LOCGPU_ERR_NO_DEVICE;
    std::memcpy(id_out, &id, sizeof(id));
    return LOCGPU_OK;
}

int locgpu_comm_init(locgpu_ctx* ctx, int rank, int world, const void* id) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(ctx, LOCGPU_ERR_INVALID, "comm_init: bad arguments");
    if (ctx->comm) return fail(ctx, LOCGPU_ERR_INVALID, "comm_init: this context already has a communicator");
    if (!rccl().ok) return fail(ctx, LOCGPU_ERR_NO_DEVICE, "comm_init: " + rccl().err);
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    // RCCL may narrow the calling thread's CPU affinity while it initialises, and threads created afterwards (the uploader's
    // packers, the tree-build pool) inherit what it leaves behind: put the caller's mask back.
    cpu_set_t saved_affinity;
    const bool have_affinity = sched_getaffinity(0, sizeof(saved_affinity), &saved_affinity) == 0;
    const ncclResult_t nr = rccl().CommInitRank(&comm, world, uid, rank);
    if (have_affinity) (void)sched_setaffinity(0, sizeof(saved_affinity), &saved_affinity);
    if (nr != ncclSuccess) return fail(ctx, LOCGPU_ERR_NO_DEVICE, std::string("ncclCommInitRank: ") + rccl().GetErrorString(nr));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return LOCGPU_OK;
}

int locgpu_comm_info(const locgpu_ctx* ctx, int* rank, int* world) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (rank) *rank = ctx->comm_rank;
    if (world) *world = ctx->comm ? ctx->comm_world : 1;
    return LOCGPU_OK;
}

void locgpu_batch_destroy(locgpu_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    (void)hipStreamSynchronize(b->stream);
    if (b->sharded && b->ctx->comm_stream) (void)hipStreamSynchronize(b->ctx->comm_stream);
    free_batch(b);
}

}  // extern "C"

// --------------------------------------------------------------------------------------------- GN driver
namespace locgpu {

static void init_states(locgpu_batch* b, const double* poses) {
    for (int s = 0; s < b->n_total; ++s) {  // an empty scan still runs the loop: effective_num < min ⇒ no-op iterations
        PoseState& ps = b->h_state[s];
        std::memset(&ps, 0, sizeof(ps));
        for (int i = 0; i < 4; ++i) ps.q[i] = poses[7 * s + i];
        for (int i = 0; i < 3; ++i) ps.t[i] = poses[7 * s + 4 + i];
        quat_to_R(ps.q, ps.R);
    }
}

static hipEvent_t get_event(locgpu_batch* b, size_t i) {
    while (b->events.size() <= i) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return nullptr;
        b->events.push_back(ev);
    }
    return b->events[i];
}

struct IterLauncher {
    locgpu_ctx* ctx;
    locgpu_batch* b;
    GnParams prm{};
    int k;
    float alpha_eff;
    size_t ev_used = 0;

    // One GN iteration = search + accumulate + solve. Returns false on a launch error.
    bool ndt = false;
    bool capturing = false;  // inside hipStreamBeginCapture: no event records
    int slot = 0;            // sharded batches: which of the chunk's exchange buffers this iteration uses
    bool replicated_on_comm_stream = false;  // the chunk's read-back must wait for the communication stream as well
    const int* active = nullptr;  // later chunks: the local scans still open (SearchArgs::active); nullptr = all
    int n_active = 0;
    const GnPost* post = nullptr;  // paced one-scan alignment: the solve kernel posts the state to the host
    bool launch(int do_update);
    void collect_profile();
};

}  // namespace locgpu

#include "ndt_inc.hpp"
#include "ndt_kernels.hpp"

namespace locgpu {

bool IterLauncher::launch(int do_update) {
    hipStream_t s = b->stream;
    const int prof = capturing ? 0 : ctx->profile;
    auto mark = [&](bool search_edge = false) {
        if (!prof || (prof == 2 && !search_edge)) return;
        hipEvent_t ev = get_event(b, ev_used);
        if (ev) { (void)hipEventRecord(ev, s); ev_used++; }
    };
    mark(true);
    int n_partial_blocks = b->blocks_per_scan;
    PoseState* st_local = b->d_state + b->first;  // kernels index the scans this rank holds: 0..n_scans-1
    if (b->n_scans == 0) {
        mark(true);  // nothing local: this rank only takes part in the exchange below
    } else if (!ndt) {
        SearchArgs sa{ctx->d_tree, ctx->tree_slots * sizeof(uint64_t), ctx->depth, batch_src(b), b->d_counts, st_local, b->d_nn, b->pitch, b->max_n, b->n_scans, k, alpha_eff,
                      prm.method == LOCGPU_P2P ? 1 : 0, ctx->count_visits ? ctx->d_visits : nullptr, b->d_redo_list, b->d_redo_count,
                      b->d_redo_list2, b->d_redo_count + 1, ctx->d_search_stats};
        const bool grid_mode = alpha_eff < 0.f && ctx->tree_bounded;
        if (alpha_eff < 0.f) sa.alpha_eff = 1.0f;                       // grid mode is exact by construction (`approximate` is ignored)
        if (!ctx->tree_bounded) sa.redo_list = nullptr;                 // huge / non-finite map coordinates: exact tree kernel only
        if (grid_mode && !b->d_grid_qkey) { fail(ctx, LOCGPU_ERR_INVALID, "grid search: work list missing (ensure_grid_lists was not called)"); return false; }
        sa.redo_list2 = b->d_redo_list2;
        sa.active = active; sa.n_active = n_active;
        if (sa.visit_totals && !capturing) {  // instrumented pass: which tree slots does this launch read at all? (bench.py: compulsory bytes)
            const size_t words = (ctx->tree_slots + 2 + 31) / 32;
            if (words > ctx->touched_words) {
                if (ctx->d_touched) (void)hipFree(ctx->d_touched);
                ctx->d_touched = nullptr; ctx->touched_words = 0;
                if (hipMalloc((void**)&ctx->d_touched, words * sizeof(uint32_t)) == hipSuccess && hipMemsetAsync(ctx->d_touched, 0, words * sizeof(uint32_t), s) == hipSuccess)
                    ctx->touched_words = words;
            }
            sa.touched = ctx->touched_words ? ctx->d_touched : nullptr;
        }
        const GridSearchScratch gsc{b->d_grid_qkey, b->d_grid_sorted, b->d_grid_tile_count, b->d_grid_scan_temp};
        const bool ok_search = (grid_mode && !sa.visit_totals) ? launch_icp_search_grid(ctx->grid, sa, gsc, s) : launch_icp_search(sa, s);
        if (!ok_search) { fail(ctx, LOCGPU_ERR_DEPTH, "search: unsupported k/depth"); return false; }
        if (sa.touched) launch_count_touched(sa.touched, (ctx->tree_slots + 2 + 31) / 32, sa.visit_totals, s);
        mark(true);
        const double gate = prm.method == LOCGPU_P2PLANE ? prm.max_plane_distance : (prm.method == LOCGPU_P2LINE ? prm.max_line_distance : prm.max_nn_distance);
        AccumArgs aa{ctx->d_tree, batch_src(b), b->d_counts, st_local, b->d_nn, b->pitch, b->max_n, b->n_scans, gate, b->d_partials};
        aa.active = active; aa.n_active = n_active;
        // a rank of a scan-sharded batch splits the partial sums as the WHOLE batch would (points per thread follow the batch's size):
        // the order of a scan's additions — hence its bits — must not depend on how many ranks share the batch (found by the eight-rank
        // loopback run of round 6: 32 of 256 scans per rank summed one point per thread where the plain batch sums four)
        if (b->sharded) aa.split_scans = b->n_total;
        n_partial_blocks = launch_icp_accum(prm.method, aa, s);
    } else {
        mark(true);  // NDT has no separate search kernel: search slot stays empty
        if (prm.method == 4)
            launch_inc_accum(ctx->inc, ctx->ndt_opts.res_outlier_th, ctx->ndt_opts.nearby_type == 0 ? 1 : 7, batch_src(b), b->d_counts, st_local,
                             b->max_n, b->n_scans, b->d_partials, s);
        else
            n_partial_blocks = launch_ndt_accum(ctx->ndt, batch_src(b), b->d_counts, st_local, b->max_n, b->n_scans, b->d_partials, s, nullptr, 0, b->sharded ? b->n_total : 0);
    }
    mark();
    if (b->sharded) {
        // The exchange step of the sharded mode: per scan 28 sums (21 H + 6 B + effective_num), zeros from the
        // ranks that do not hold the scan, summed over xGMI on this stream; then every rank solves every scan, so all ranks see the
        // same convergence flags and stay in lock-step.
        double* acc = b->d_acc + (size_t)(slot % kFirstChunk) * b->n_total * kAccW;
        slot++;
        launch_sum_partials(b->d_partials, n_partial_blocks, b->d_state, b->first, b->n_scans, b->n_total, acc, s);
        // Scan-sharded over several ranks: a scan's sums are complete on the rank that holds it (everybody else adds zeros), so the
        // OWNER solves its scans at once and goes on to the next search, while the all-reduce — on the context's communication
        // stream — only replicates: behind it every rank solves the scans it does not hold, from the reduced sums, and ends up with
        // the same poses and flags as their owners. The network is off the Gauss–Newton loop's critical path; the host looks at the
        // flags of ALL scans only between chunks (both streams joined), which keeps the ranks' collective counts in lock-step.
        // Point-sharded batches (every rank holds a slice of every scan) and H/B evaluations need the sum itself: they wait.
        static const int decouple_env = [] { const char* e = getenv("LOCGPU_SHARD_DECOUPLED"); return e ? atoi(e) : -1; }();
        const bool scan_sharded = b->n_scans != b->n_total;
        const bool decoupled = ctx->comm && do_update && scan_sharded && (decouple_env >= 0 ? decouple_env != 0 : ctx->comm_world > 1);
        if (decoupled) {
            hipStream_t cs = ctx->comm_stream;
            if (b->n_scans > 0)
                launch_gn_solve(acc + (size_t)b->first * kAccW, 1, b->d_state + b->first, b->n_scans, prm, do_update, b->d_hb + (size_t)b->first * 44,
                                ndt ? nullptr : b->d_redo_count, s);
            if (!hip_ok(ctx, hipEventRecord(b->ev_ready, s), "sharded: hipEventRecord") || !hip_ok(ctx, hipStreamWaitEvent(cs, b->ev_ready, 0), "sharded: hipStreamWaitEvent")) return false;
            const ncclResult_t nr = rccl().AllReduce(acc, acc, (size_t)b->n_total * kAccW, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, cs);
            if (nr != ncclSuccess) { fail(ctx, LOCGPU_ERR_NO_DEVICE, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr)); return false; }
            const int after = b->first + b->n_scans;
            if (b->first > 0) launch_gn_solve(acc, 1, b->d_state, b->first, prm, do_update, b->d_hb, nullptr, cs);
            if (after < b->n_total)
                launch_gn_solve(acc + (size_t)after * kAccW, 1, b->d_state + after, b->n_total - after, prm, do_update, b->d_hb + (size_t)after * 44, nullptr, cs);
            replicated_on_comm_stream = true;
            mark();
            return hip_ok(ctx, hipGetLastError(), "kernel launch");
        }
        if (ctx->comm) {
            // Every collective of the context goes through ONE stream in host order — the order is the same on every rank because
            // every rank sees the same convergence flags — so two batches in flight never have two collectives of the one
            // communicator racing each other.
            // (A one-rank communicator has nobody to disagree with about the order: its collective stays on the batch's own stream —
            // 32 scans per step, two in flight: 8500 scans/s against 5700 through the comm stream, whose in-order queue makes the
            // second batch's first exchange wait for the first batch's whole chunk. LOCGPU_COMM_DIRECT=0/1 forces either way.)
            static const int force = [] { const char* e = getenv("LOCGPU_COMM_DIRECT"); return e ? atoi(e) : -1; }();
            const bool direct = force >= 0 ? force != 0 : ctx->comm_world == 1;
            hipStream_t cs = direct ? s : ctx->comm_stream;
            if (!direct && (!hip_ok(ctx, hipEventRecord(b->ev_ready, s), "sharded: hipEventRecord") || !hip_ok(ctx, hipStreamWaitEvent(cs, b->ev_ready, 0), "sharded: hipStreamWaitEvent"))) return false;
            const ncclResult_t nr = rccl().AllReduce(acc, acc, (size_t)b->n_total * kAccW, ncclDouble, ncclSum, (ncclComm_t)ctx->comm, cs);
            if (nr != ncclSuccess) { fail(ctx, LOCGPU_ERR_NO_DEVICE, std::string("ncclAllReduce: ") + rccl().GetErrorString(nr)); return false; }
            if (!direct && (!hip_ok(ctx, hipEventRecord(b->ev_reduced, cs), "sharded: hipEventRecord") || !hip_ok(ctx, hipStreamWaitEvent(s, b->ev_reduced, 0), "sharded: hipStreamWaitEvent"))) return false;
        }
        launch_gn_solve(acc, 1, b->d_state, b->n_total, prm, do_update, b->d_hb, ndt ? nullptr : b->d_redo_count, s);
    } else {
        launch_gn_solve(b->d_partials, n_partial_blocks, b->d_state, b->n_scans, prm, do_update, b->d_hb, ndt ? nullptr : b->d_redo_count, s, nullptr, post);
    }
    mark();
    return hip_ok(ctx, hipGetLastError(), "kernel launch");
}

void IterLauncher::collect_profile() {
    // launch() records four events per iteration: [0,1] search, [1,2] fit+accumulate, [2,3] solve — or, in the light mode, two: [0,1] search.
    if (ctx->profile == 2) {
        for (size_t i = 0; i + 1 < ev_used; i += 2) {
            float ms = 0.f;
            if (!ndt && hipEventElapsedTime(&ms, b->events[i], b->events[i + 1]) == hipSuccess) {
                ctx->prof_ms[0] += ms;
                ctx->prof_n[0] += 1;
            }
        }
    } else if (ctx->profile) {
        for (size_t i = 0; i + 3 < ev_used; i += 4)
            for (int j = 0; j < 3; ++j) {
                if (ndt && j == 0) continue;  // NDT has no search kernel
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, b->events[i + j], b->events[i + j + 1]) == hipSuccess) {
                    ctx->prof_ms[j] += ms;
                    ctx->prof_n[j] += 1;
                }
            }
    }
    ev_used = 0;
}

static void write_results(locgpu_batch* b, const double* init_poses, double* out_poses, locgpu_align_stats* stats);

// The grid search hands its leftovers through a second work list. It is allocated here, by every entry point that may run the
// grid search on `b`, BEFORE any launch: launch() can run under hipStreamBeginCapture, where hipMalloc is not allowed.
// A pending locgpu_batch_upload_async of `b`: wait until the host side is through, then order the compute stream behind the copies.
static int batch_ready(locgpu_ctx* ctx, locgpu_batch* b) {
    const int rc = upload_join_batch(b);
    if (rc != LOCGPU_OK) return rc;
    LOCGPU_HIP(ctx, upload_order_after(b, b->stream));
    return LOCGPU_OK;
}

static int ensure_grid_lists(locgpu_ctx* ctx, locgpu_batch* b, float alpha_eff) {
    if (alpha_eff >= 0.f) return LOCGPU_OK;
    if (!b->d_grid_qkey) {
        LOCGPU_HIP(ctx, hipMalloc((void**)&b->d_grid_qkey, b->pitch * sizeof(uint32_t)));
        LOCGPU_HIP(ctx, hipMalloc((void**)&b->d_grid_sorted, b->pitch * sizeof(uint2)));
    }
    // the binning's per-tile counts and scan workspace are the batch's own as well (several alignments run at once); sized by the
    // current target's grid — a new target may have more occupied tiles
    const size_t tocc = ctx->grid.n_tocc, scan = std::max<size_t>(ctx->grid.scan_temp_bytes, 1);
    if (!b->d_grid_tile_count || b->grid_tocc_cap < tocc) {
        LOCGPU_HIP(ctx, hipStreamSynchronize(b->stream));
        if (b->d_grid_tile_count) (void)hipFree(b->d_grid_tile_count);
        b->d_grid_tile_count = nullptr;
        LOCGPU_HIP(ctx, hipMalloc((void**)&b->d_grid_tile_count, (tocc + 1) * sizeof(uint32_t)));
        b->grid_tocc_cap = tocc;
    }
    if (!b->d_grid_scan_temp || b->grid_scan_cap < scan) {
        LOCGPU_HIP(ctx, hipStreamSynchronize(b->stream));
        if (b->d_grid_scan_temp) (void)hipFree(b->d_grid_scan_temp);
        b->d_grid_scan_temp = nullptr;
        LOCGPU_HIP(ctx, hipMalloc(&b->d_grid_scan_temp, scan));
        b->grid_scan_cap = scan;
    }
    return LOCGPU_OK;
}

// hipGraph path (BASELINE config 5): the Gauss–Newton iterations are captured once — the kernels early-out per scan on the
// device-side `done` flag, so a fixed node sequence gives the same result as the data-dependent eager loop — and replayed per call.
// Two graphs mirror the eager loop's chunks: graph 0 = {H2D state, kFirstChunk iterations, D2H state} covers the typical alignment
// with one launch and one host synchronisation; graph 1 = {kNextChunk iterations, D2H state} is replayed while scans are still
// open (capturing all max_iteration iterations in one graph made every call pay a dozen empty iterations).
static int capture_chunk(locgpu_ctx* ctx, locgpu_batch* b, const GnParams& prm, int k, float alpha_eff, bool ndt, int iters, bool with_h2d,
                         hipGraphExec_t* out) {
    hipStream_t s = b->stream;
    hipGraph_t graph = nullptr;
    LOCGPU_HIP(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    bool ok = !with_h2d || hip_ok(ctx, hipMemcpyAsync(b->d_state, b->h_state, b->n_total * sizeof(PoseState), hipMemcpyHostToDevice, s), "capture H2D");
    IterLauncher it{ctx, b, prm, k, alpha_eff};
    it.ndt = ndt;
    it.capturing = true;
    for (int i = 0; ok && i < iters; ++i) ok = it.launch(1);
    ok = ok && hip_ok(ctx, hipMemcpyAsync(b->h_state, b->d_state, b->n_total * sizeof(PoseState), hipMemcpyDeviceToHost, s), "capture D2H");
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (!ok || !hip_ok(ctx, e, "hipStreamEndCapture")) { if (graph) (void)hipGraphDestroy(graph); return LOCGPU_ERR_NO_DEVICE; }
    const bool inst = hip_ok(ctx, hipGraphInstantiate(out, graph, nullptr, nullptr, 0), "hipGraphInstantiate");
    (void)hipGraphDestroy(graph);
    if (!inst) { *out = nullptr; return LOCGPU_ERR_NO_DEVICE; }
    return LOCGPU_OK;
}

// An alignment in two halves, so that a caller can have two batches in flight (their streams differ): align_begin enqueues the
// first chunk of iterations and returns; align_finish waits for it, enqueues further chunks while scans are still open, and
// writes the results. The blocking entry points are begin + finish back to back.
static int ensure_graphs(locgpu_ctx* ctx, locgpu_batch* b, const GnParams& prm, int k, float alpha_eff, bool ndt) {
    const void* target = !ndt ? (const void*)ctx->d_tree : (prm.method == 4 ? inc_ndt_table_ptr(ctx->inc) : (const void*)ctx->ndt->d_rec);
    const bool same = b->graph_exec && b->graph_k == k && b->graph_alpha == alpha_eff && b->graph_ndt == ndt && b->graph_target == target &&
                      b->graph_epoch == ctx->target_epoch &&
                      b->graph_prm == prm;
    if (same) return LOCGPU_OK;
    const int first = std::min(kFirstChunk, prm.max_iteration);
    if (b->graph_exec) { (void)hipGraphExecDestroy(b->graph_exec); b->graph_exec = nullptr; }
    if (b->graph_exec_next) { (void)hipGraphExecDestroy(b->graph_exec_next); b->graph_exec_next = nullptr; }
    int rc = capture_chunk(ctx, b, prm, k, alpha_eff, ndt, first, true, &b->graph_exec);
    if (rc == LOCGPU_OK && prm.max_iteration > first) rc = capture_chunk(ctx, b, prm, k, alpha_eff, ndt, next_chunk(b), false, &b->graph_exec_next);
    if (rc != LOCGPU_OK) return rc;
    b->graph_prm = prm; b->graph_k = k; b->graph_alpha = alpha_eff; b->graph_ndt = ndt; b->graph_target = target;
    b->graph_epoch = ctx->target_epoch;
    return LOCGPU_OK;
}

// One chunk of iterations + the read-back of the per-scan states behind it, on the batch's stream.
static int enqueue_chunk(locgpu_ctx* ctx, locgpu_batch* b, bool first_chunk) {
    locgpu_batch::Pending& P = b->pending;
    hipStream_t s = b->stream;
    if (P.graph) {
        // kernels of a finished scan return at once and the solve kernel stops at max_iteration, so a whole chunk is always safe
        LOCGPU_HIP(ctx, hipGraphLaunch(first_chunk ? b->graph_exec : b->graph_exec_next, s));
        P.launched += first_chunk ? std::min(kFirstChunk, P.prm.max_iteration) : next_chunk(b);
        return LOCGPU_OK;
    }
    if (first_chunk) LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_state, b->h_state, b->n_total * sizeof(PoseState), hipMemcpyHostToDevice, s));
    IterLauncher it{ctx, b, P.prm, P.k, P.alpha_eff};
    it.ndt = P.ndt;
    it.ev_used = P.ev_used;
    if (!first_chunk && !P.ndt && b->n_scans > 1) {
        // The host has just read every scan's flags (align_finish): launch the search and accumulate kernels of this chunk over the
        // local scans still open only. A 256-scan step's second and third chunk hold ≈60 and ≈5 scans; the rest used to be 1800
        // early-exit workgroups per scan and kernel (≈96 µs per search launch for nothing). Results are the same bits: a scan's
        // blocks do the same work wherever blockIdx.y finds it, and the accumulate kernels' split does not depend on the list.
        int na = 0;
        for (int i = 0; i < b->n_scans; ++i)
            if (!b->h_state[b->first + i].done) b->h_active[na++] = i;
        if (na > 0 && na < b->n_scans) {
            LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_active, b->h_active, (size_t)na * sizeof(int), hipMemcpyHostToDevice, s));
            it.active = b->d_active;
            it.n_active = na;
        }
    }
    const int todo = std::min(first_chunk ? first_chunk_len(b) : next_chunk(b), P.prm.max_iteration - P.launched);
    for (int c = 0; c < todo; ++c)
        if (!it.launch(1)) return LOCGPU_ERR_NO_DEVICE;
    P.ev_used = it.ev_used;
    P.launched += todo;
    if (it.replicated_on_comm_stream) {  // the states of the scans other ranks hold are written on the communication stream
        LOCGPU_HIP(ctx, hipEventRecord(b->ev_reduced, ctx->comm_stream));
        LOCGPU_HIP(ctx, hipStreamWaitEvent(s, b->ev_reduced, 0));
    }
    LOCGPU_HIP(ctx, hipMemcpyAsync(b->h_state, b->d_state, b->n_total * sizeof(PoseState), hipMemcpyDeviceToHost, s));
    return LOCGPU_OK;
}

// A ONE-SCAN alignment is paced from the host instead of chunked: the solve kernel posts the scan's state and an iteration word to
// pinned host memory (GnPost, icp_kernels.hip); the host keeps `ahead` iterations queued behind the one that is running and launches
// the next when a post arrives. Against chunks (first_chunk_len / next_chunk above, still what graphs and batches use) a call no
// longer pays the idle iterations of a chunk that was sized by the previous call (≈14 µs each: three dispatches that find `done`),
// nor a chunk boundary (read-back + host + relaunch ≈ 33 µs) when the guess was short, nor the copy and the stream synchronisation
// at the end: the result is in host memory when the done bit arrives. At most `ahead` idle iterations stay queued behind a finished
// call; they return on the `done` flag before they read anything (an upload or the next call's state copy may follow at once).
// Same kernels on the same data in the same order: results are the chunked path's bits. LOCGPU_PACE_AHEAD=0 switches it off.
inline int pace_ahead() {
    static const int v = [] { const char* e = getenv("LOCGPU_PACE_AHEAD"); return e ? std::max(0, std::min(8, atoi(e))) : 1; }();
    return v;
}

static int paced_launch(locgpu_ctx* ctx, locgpu_batch* b, int upto) {
    locgpu_batch::Pending& P = b->pending;
    IterLauncher it{ctx, b, P.prm, P.k, P.alpha_eff};
    it.ndt = P.ndt;
    const GnPost post{reinterpret_cast<GnPostRecord*>(b->h_post), b->h_post + locgpu_batch::kPostWord, b->post_call};
    it.post = &post;
    while (P.launched < upto) {
        if (!it.launch(1)) return LOCGPU_ERR_NO_DEVICE;
        P.launched++;
    }
    return LOCGPU_OK;
}

// Wait for a post of this call that is newer than iteration `seen`; returns the word. A post with the done bit is taken only when the
// state behind it is complete (its checksum matches what this thread reads: the kernel's stores carry no fence).
constexpr int kPacedTimeoutS = 30;  // an iteration of the largest alignment this path takes (one scan) lasts well under a millisecond
static int paced_wait(locgpu_ctx* ctx, locgpu_batch* b, int seen, unsigned long long* out) {
    auto fresh = [&](unsigned long long* w_out) {
        unsigned long long w;
        GnPostRecord r;
        if (!gn_post_take(b->h_post, locgpu_batch::kPostWord, b->post_call, seen, &w, &r)) return false;
        if (w & 1ull) {
            PoseState& ps = b->h_state[0];
            for (int i = 0; i < 4; ++i) std::memcpy(&ps.q[i], &r.w[i], 8);
            for (int i = 0; i < 3; ++i) std::memcpy(&ps.t[i], &r.w[4 + i], 8);
            quat_to_R(ps.q, ps.R);
            std::memcpy(&ps.last_dx_norm, &r.w[7], 8);
            ps.last_eff = (long long)r.w[8];
            ps.iterations = (int)((w & 0xffffffffull) >> 1);
            ps.converged = (int)(r.w[9] >> 32);
            ps.status = (int)(r.w[9] & 0xffffffffull);
            ps.done = 1;
        }
        *w_out = w;
        return true;
    };
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned long spins = 1;; ++spins) {
        if (fresh(out)) return LOCGPU_OK;
#if defined(__x86_64__)
        __builtin_ia32_pause();  // a polite spin: the sibling hyper-thread (the uploader, the helper thread) gets the core's issue slots
#endif
        if ((spins & 0xffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
            // nothing for a long time: is the stream still working? An idle stream with no post means a kernel died or the posts do
            // not reach the host. (Not earlier: a stream query is a runtime call on the latency path.)
            const hipError_t q = hipStreamQuery(b->stream);
            if (q == hipErrorNotReady) {
                // a stream that stays busy without ever posting (a hung kernel) must not spin a core for ever: give up after kPacedTimeoutS
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(kPacedTimeoutS))
                    return fail(ctx, LOCGPU_ERR_NO_DEVICE, "paced alignment: no post from the solve kernel within the time-out (the stream is still busy)");
                std::this_thread::yield();
                continue;
            }
            if (q != hipSuccess) { hip_ok(ctx, q, "paced alignment"); return LOCGPU_ERR_NO_DEVICE; }
            if (fresh(out)) return LOCGPU_OK;
            return fail(ctx, LOCGPU_ERR_NO_DEVICE, "paced alignment: the stream is idle and the solve kernel's post has not arrived");
        }
    }
}

static int align_begin(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses, const GnParams& prm, int k, float alpha_eff, bool ndt,
                       bool blocking = false) {
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    locgpu_batch::Pending& P = b->pending;
    if (P.active) return fail(ctx, LOCGPU_ERR_INVALID, "align: an alignment of this batch has been begun and not finished");
    if (!ndt) { const int grc = ensure_grid_lists(ctx, b, alpha_eff); if (grc != LOCGPU_OK) return grc; }
    { const int urc = batch_ready(ctx, b); if (urc != LOCGPU_OK) return urc; }
    P.prm = prm; P.k = k; P.alpha_eff = alpha_eff; P.ndt = ndt;
    P.graph = ctx->use_graph && !ctx->count_visits && !b->sharded && prm.max_iteration > 0;
    P.launched = 0;
    P.ev_used = 0;
    P.init_poses.assign(init_poses, init_poses + 7 * (size_t)b->n_total);
    init_states(b, init_poses);
    // the search stage's work-list counters: zero once per alignment, whatever an earlier call that failed between a search and
    // its solve kernel left behind (the solve kernel re-zeroes them after every search)
    if (!ndt && !b->counters_clean) LOCGPU_HIP(ctx, hipMemsetAsync(b->d_redo_count, 0, 4 * sizeof(unsigned int), b->stream));
    b->counters_clean = false;  // until this alignment has run to its end
    if (P.graph) { const int rc = ensure_graphs(ctx, b, prm, k, alpha_eff, ndt); if (rc != LOCGPU_OK) return rc; }
    // (a blocking call only: between a begin and its end the host is elsewhere, and a chunk keeps the GPU busy meanwhile)
    P.paced = blocking && !P.graph && b->n_total == 1 && !b->sharded && !ctx->profile && !ctx->count_visits && prm.max_iteration > 0 && pace_ahead() > 0 && (ndt || alpha_eff >= 0.f);
    if (P.paced) {
        if (!b->h_post) {
            LOCGPU_HIP(ctx, hipHostMalloc((void**)&b->h_post, 256, hipHostMallocCoherent));
            std::memset(b->h_post, 0, 256);
        }
        b->post_call++;  // posts carry the call's number: a word left by the previous call is not this call's
        LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_state, b->h_state, sizeof(PoseState), hipMemcpyHostToDevice, b->stream));
        const int rc = paced_launch(ctx, b, std::min(prm.max_iteration, 1 + pace_ahead()));
        if (rc != LOCGPU_OK) { (void)hipStreamSynchronize(b->stream); return rc; }
    } else if (prm.max_iteration > 0) {
        const int rc = enqueue_chunk(ctx, b, true);
        if (rc != LOCGPU_OK) return rc;
    }
    P.active = true;
    return LOCGPU_OK;
}

static int align_finish(locgpu_ctx* ctx, locgpu_batch* b, double* out_poses, locgpu_align_stats* stats) {
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    locgpu_batch::Pending& P = b->pending;
    if (!P.active) return fail(ctx, LOCGPU_ERR_INVALID, "align: no alignment of this batch has been begun");
    P.active = false;
    if (P.paced) {
        int seen = 0;
        for (;;) {
            unsigned long long w = 0;
            int rc = paced_wait(ctx, b, seen, &w);
            if (rc == LOCGPU_OK) {
                seen = (int)((w & 0xffffffffull) >> 1);
                if ((w & 1ull) || seen >= P.prm.max_iteration) break;
                rc = paced_launch(ctx, b, std::min(P.prm.max_iteration, seen + 1 + pace_ahead()));
            }
            if (rc != LOCGPU_OK) { (void)hipStreamSynchronize(b->stream); return rc; }
        }
        b->paced_tail = true;
    }
    while (!P.paced && P.prm.max_iteration > 0) {
        LOCGPU_HIP(ctx, hipStreamSynchronize(b->stream));
        if (!P.graph) {
            IterLauncher it{ctx, b, P.prm, P.k, P.alpha_eff};
            it.ndt = P.ndt;
            it.ev_used = P.ev_used;
            it.collect_profile();
            P.ev_used = 0;
        }
        bool all_done = true;
        for (int i = 0; i < b->n_total; ++i)
            if (!b->h_state[i].done) { all_done = false; break; }
        if (all_done || P.launched >= P.prm.max_iteration) break;
        const int rc = enqueue_chunk(ctx, b, false);
        if (rc != LOCGPU_OK) {
            // whatever of the chunk was enqueued must not run on under the batch's next upload (which relies on an ended alignment
            // leaving its stream idle, batch_upload.hip)
            (void)hipStreamSynchronize(b->stream);
            return rc;
        }
    }
    write_results(b, P.init_poses.data(), out_poses, stats);
    b->counters_clean = !P.ndt && !ctx->count_visits && P.alpha_eff >= 0.f && !b->sharded;  // every search was followed by its solve kernel, which zeroes them (a one-scan front-end saves a fill launch per call)
    if (b->n_total == 1 && P.prm.max_iteration > 0) b->last_iterations = b->h_state[0].iterations;
    return LOCGPU_OK;
}

static int run_align(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses, const GnParams& prm, int k, float alpha_eff, bool ndt,
                     double* out_poses, locgpu_align_stats* stats) {
    const int rc = align_begin(ctx, b, init_poses, prm, k, alpha_eff, ndt, /*blocking*/ true);
    return rc != LOCGPU_OK ? rc : align_finish(ctx, b, out_poses, stats);
}

static void write_results(locgpu_batch* b, const double* init_poses, double* out_poses, locgpu_align_stats* stats) {
    for (int i = 0; i < b->n_total; ++i) {
        const PoseState& ps = b->h_state[i];
        if (ps.status == 1) {  // direct NDT aborted: reference leaves result_pose unassigned; hand back init_pose
            for (int j = 0; j < 7; ++j) out_poses[7 * i + j] = init_poses[7 * i + j];
        } else {
            for (int j = 0; j < 4; ++j) out_poses[7 * i + j] = ps.q[j];
            for (int j = 0; j < 3; ++j) out_poses[7 * i + 4 + j] = ps.t[j];
        }
        if (stats) {
            stats[i].iterations = ps.iterations;
            stats[i].converged = ps.converged;
            stats[i].status = ps.status;
            stats[i].reserved = 0;
            stats[i].last_effective_num = ps.last_eff;
            stats[i].last_dx_norm = ps.last_dx_norm;
        }
    }
}

int check_icp(locgpu_ctx* ctx, const locgpu_icp_opts* o, GnParams& prm, int& k, float& alpha_eff) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!o) return fail(ctx, LOCGPU_ERR_INVALID, "icp: opts is NULL");
    { const int jrc = target_join(ctx); if (jrc != LOCGPU_OK) return jrc; }  // an asynchronous SetInputTarget ends here at the latest
    if (!ctx->d_tree) return fail(ctx, LOCGPU_ERR_NO_TARGET, "icp: SetInputTarget has not been called");
    if (o->method < LOCGPU_P2P || o->method > LOCGPU_P2PLANE) return fail(ctx, LOCGPU_ERR_INVALID, "icp: unknown method");
    if (o->search_mode != LOCGPU_SEARCH_TREE_FAITHFUL && o->search_mode != LOCGPU_SEARCH_GRID_EXACT)
        return fail(ctx, LOCGPU_ERR_INVALID, "icp: unknown search mode");
    if (o->search_mode == LOCGPU_SEARCH_GRID_EXACT) {
        const int rc = ensure_grid(ctx);
        if (rc != LOCGPU_OK) return rc;
    }
    prm.method = o->method;
    prm.max_iteration = o->max_iteration;
    prm.min_effective_pts = o->min_effective_pts;
    prm.eps = o->eps;
    prm.max_nn_distance = o->max_nn_distance;
    prm.max_plane_distance = o->max_plane_distance;
    prm.max_line_distance = o->max_line_distance;
    k = o->method == LOCGPU_P2P ? 1 : 5;
    alpha_eff = o->approximate ? o->ann_alpha : 1.0f;
    if (o->search_mode == LOCGPU_SEARCH_GRID_EXACT) alpha_eff = -1.0f;  // marker: grid search (exact by construction; `approximate` is ignored)
    // k > size_: GetClosestPoint logs an error and returns nothing (kdtree.cpp:149-153) ⇒ no correspondences at all.
    // The search kernel reproduces that by never filling the k-th slot; nothing to reject here.
    return LOCGPU_OK;
}

// The reusable one-scan batch (device buffers + pinned staging) with room for n points.
static int single_reserve(locgpu_ctx* ctx, size_t n, locgpu_batch** out) {
    if (n == 0) return fail(ctx, LOCGPU_ERR_INVALID, "source cloud is empty");
    locgpu_batch* b = ctx->single;
    if (!b || (size_t)b->max_n < n || !b->h_src) {
        if (b) { free_batch(b); ctx->single = nullptr; }
        // device buffers with headroom, then pinned staging of the same capacity
        const size_t cap = n + n / 4 + 1024;
        const int rc = alloc_batch(ctx, 1, cap, &ctx->single);
        if (rc != LOCGPU_OK) return rc;
        b = ctx->single;
        b->slot = 0;  // single-scan calls share the stream of the clouds and of the target ingest
        b->stream = ctx->stream;
        if (!hip_ok(ctx, hipHostMalloc((void**)&b->h_src, cap * sizeof(float4)), "hipHostMalloc src")) { free_batch(b); ctx->single = nullptr; return LOCGPU_ERR_OOM; }
    }
    *out = b;
    return LOCGPU_OK;
}

static int single_batch(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, locgpu_batch** out) {
    // The reference deep-copies the source on every call (SetSource, icp_registration.cpp:252-265); so do we — but into buffers
    // that are kept between calls: a per-scan caller (Loc::Update at 10-20 Hz) must not pay a dozen hipMalloc/hipFree per scan.
    if (n == 0) return fail(ctx, LOCGPU_ERR_INVALID, "source cloud is empty");
    if (!src || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "source cloud: NULL pointer or stride < 12");
    locgpu_batch* b = nullptr;
    const int rc = single_reserve(ctx, n, &b);
    if (rc != LOCGPU_OK) return rc;
    pack_points((const char*)src, stride_bytes, n, b->h_src);
    b->d_src_ext = nullptr;
    b->counts[0] = (int)n;
    LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_src, b->h_src, n * sizeof(float4), hipMemcpyHostToDevice, b->stream));
    LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_counts, b->counts.data(), sizeof(int), hipMemcpyHostToDevice, b->stream));
    *out = b;
    return LOCGPU_OK;
}

// Same, from a cloud that is already in HBM (the w lane carries the intensity; no kernel of the matcher reads it).
static int single_batch_dev(locgpu_ctx* ctx, const float4* d_src, size_t n, locgpu_batch** out) {
    locgpu_batch* b = nullptr;
    const int rc = single_reserve(ctx, n, &b);
    if (rc != LOCGPU_OK) return rc;
    b->counts[0] = (int)n;
    // The call that follows is synchronous and only reads the points: they stay where the cloud holds them (a D2D copy was one more
    // dispatch in front of every match of the streaming loop). A captured graph has the batch's own buffer in its kernel arguments.
    b->d_src_ext = ctx->use_graph ? nullptr : d_src;
    if (!b->d_src_ext) LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_src, d_src, n * sizeof(float4), hipMemcpyDeviceToDevice, b->stream));
    LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_counts, b->counts.data(), sizeof(int), hipMemcpyHostToDevice, b->stream));
    *out = b;
    return LOCGPU_OK;
}

}  // namespace locgpu

extern "C" {

int locgpu_icp_align_batch(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses, const locgpu_icp_opts* opts, double* out_poses,
                           locgpu_align_stats* stats) {
    GnParams prm{};
    int k;
    float alpha_eff;
    const int rc = check_icp(ctx, opts, prm, k, alpha_eff);
    if (rc != LOCGPU_OK) return rc;
    if (!b || b->ctx != ctx || !init_poses || !out_poses) return fail(ctx, LOCGPU_ERR_INVALID, "icp_align_batch: bad arguments");
    return run_align(ctx, b, init_poses, prm, k, alpha_eff, false, out_poses, stats);
}

int locgpu_icp_align_batch_begin(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses, const locgpu_icp_opts* opts) {
    GnParams prm{};
    int k;
    float alpha_eff;
    const int rc = check_icp(ctx, opts, prm, k, alpha_eff);
    if (rc != LOCGPU_OK) return rc;
    if (!b || b->ctx != ctx || !init_poses) return fail(ctx, LOCGPU_ERR_INVALID, "icp_align_batch_begin: bad arguments");
    return align_begin(ctx, b, init_poses, prm, k, alpha_eff, false);
}

int locgpu_align_batch_end(locgpu_ctx* ctx, locgpu_batch* b, double* out_poses, locgpu_align_stats* stats) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!b || b->ctx != ctx || !out_poses) return fail(ctx, LOCGPU_ERR_INVALID, "align_batch_end: bad arguments");
    return align_finish(ctx, b, out_poses, stats);
}

int locgpu_icp_align(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double init_pose[7], const locgpu_icp_opts* opts,
                     double out_pose[7], locgpu_align_stats* stats) {
    GnParams prm{};
    int k;
    float alpha_eff;
    int rc = check_icp(ctx, opts, prm, k, alpha_eff);
    if (rc != LOCGPU_OK) return rc;
    if (!src || !init_pose || !out_pose) return fail(ctx, LOCGPU_ERR_INVALID, "icp_align: bad arguments");
    locgpu_batch* b = nullptr;
    rc = single_batch(ctx, src, n, stride_bytes, &b);
    if (rc != LOCGPU_OK) return rc;
    return run_align(ctx, b, init_pose, prm, k, alpha_eff, false, out_pose, stats);
}

int locgpu_icp_hb_batch(locgpu_ctx* ctx, locgpu_batch* b, const double* poses, const locgpu_icp_opts* opts, double* hb) {
    GnParams prm{};
    int k;
    float alpha_eff;
    const int rc = check_icp(ctx, opts, prm, k, alpha_eff);
    if (rc != LOCGPU_OK) return rc;
    if (!b || b->ctx != ctx || !poses || !hb) return fail(ctx, LOCGPU_ERR_INVALID, "icp_hb_batch: bad arguments");
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    { const int grc = ensure_grid_lists(ctx, b, alpha_eff); if (grc != LOCGPU_OK) return grc; }
    { const int urc = batch_ready(ctx, b); if (urc != LOCGPU_OK) return urc; }
    if (b->pending.active) return fail(ctx, LOCGPU_ERR_INVALID, "icp_hb_batch: an alignment of this batch has been begun and not finished");
    init_states(b, poses);
    b->counters_clean = false;
    LOCGPU_HIP(ctx, hipMemsetAsync(b->d_redo_count, 0, 4 * sizeof(unsigned int), b->stream));
    LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_state, b->h_state, b->n_total * sizeof(PoseState), hipMemcpyHostToDevice, b->stream));
    IterLauncher it{ctx, b, prm, k, alpha_eff};
    if (!it.launch(0)) return LOCGPU_ERR_NO_DEVICE;
    LOCGPU_HIP(ctx, hipMemcpyAsync(b->h_hb, b->d_hb, (size_t)b->n_total * 44 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    LOCGPU_HIP(ctx, hipStreamSynchronize(b->stream));
    it.collect_profile();
    std::memcpy(hb, b->h_hb, (size_t)b->n_total * 44 * sizeof(double));
    return LOCGPU_OK;
}

int locgpu_debug_batch_nn(locgpu_ctx* ctx, locgpu_batch* b, int k, int32_t* out) {
    if (!ctx || !b || b->ctx != ctx || !out || k < 1 || k > 5) return fail(ctx, LOCGPU_ERR_INVALID, "debug_batch_nn: bad arguments");
    { const int jrc = target_join(ctx); if (jrc != LOCGPU_OK) return jrc; }
    if (!ctx->d_tree) return fail(ctx, LOCGPU_ERR_NO_TARGET, "debug_batch_nn: no ICP target");
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    const size_t nq = (size_t)b->n_scans * b->max_n;
    int32_t* d_out = nullptr;
    LOCGPU_HIP(ctx, hipMalloc((void**)&d_out, nq * k * sizeof(int32_t)));
    launch_nn_to_index(ctx->d_tree, b->d_nn, b->pitch, nq, k, d_out, b->stream);
    const hipError_t e = hipMemcpyAsync(out, d_out, nq * k * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream);
    const hipError_t e2 = hipStreamSynchronize(b->stream);
    (void)hipFree(d_out);
    LOCGPU_HIP(ctx, e);
    LOCGPU_HIP(ctx, e2);
    return LOCGPU_OK;
}

int locgpu_icp_hb(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double pose[7], const locgpu_icp_opts* opts, double H[36],
                  double B[6], int64_t* effective_num, int* ok) {
    GnParams prm{};
    int k;
    float alpha_eff;
    int rc = check_icp(ctx, opts, prm, k, alpha_eff);
    if (rc != LOCGPU_OK) return rc;
    if (!src || !pose || !H || !B) return fail(ctx, LOCGPU_ERR_INVALID, "icp_hb: bad arguments");
    locgpu_batch* b = nullptr;
    rc = single_batch(ctx, src, n, stride_bytes, &b);
    if (rc != LOCGPU_OK) return rc;
    double hb[44];
    rc = locgpu_icp_hb_batch(ctx, b, pose, opts, hb);
    if (rc != LOCGPU_OK) return rc;
    std::memcpy(H, hb, 36 * sizeof(double));
    std::memcpy(B, hb + 36, 6 * sizeof(double));
    if (effective_num) *effective_num = (int64_t)hb[42];
    if (ok) *ok = hb[43] != 0.0;
    return LOCGPU_OK;
}

int locgpu_gn_update(const double hb[44], int method, int min_effective_pts, double eps, double pose[7], double dx[6], int* applied, int* stop) {
    if (!hb || !pose || !dx) return LOCGPU_ERR_INVALID;
    for (int i = 0; i < 6; ++i) dx[i] = 0.0;
    if (applied) *applied = 0;
    if (stop) *stop = 0;
    const double det = lu6_det_solve(hb, hb + 36, dx);
    const bool ok = ((long long)hb[42] >= min_effective_pts) && !(det == 0.0);
    if (!ok) { for (int i = 0; i < 6; ++i) dx[i] = 0.0; return LOCGPU_OK; }
    if (method == LOCGPU_P2P)
        for (int i = 0; i < 6; ++i) dx[i] = dx[i] / 16;
    se3_apply_update(pose, pose + 4, dx);
    double n2 = 0.0;
    for (int i = 0; i < 6; ++i) n2 += dx[i] * dx[i];
    if (applied) *applied = 1;
    if (stop) *stop = std::sqrt(n2) < eps;
    return LOCGPU_OK;
}

// The output cloud of ScanMatch (pcl::transformPointCloud(*src, *out, pose.matrix().cast<float>()), icp_registration.cpp:241,
// ndt_registration.cpp:258): the points of the one-scan batch `b` (they are in HBM from the alignment, or were just uploaded) go through
// transform_cloud_kernel into the batch's neighbour-list array (dead by now: 20 B per point, 12 needed), come back through the batch's
// pinned staging and are written x, y, z into the caller's points. No allocation, no second upload, one kernel, one copy.
static int write_output_cloud(locgpu_ctx* ctx, locgpu_batch* b, const float4* d_points, size_t n, const double pose[7], void* out, size_t out_stride) {
    double R[9];
    quat_to_R(pose, R);
    M12f m;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m.v[4 * r + c] = (float)R[3 * r + c];
        m.v[4 * r + 3] = (float)pose[4 + r];
    }
    float* d_xyz = reinterpret_cast<float*>(b->d_nn);
    float* h_xyz = reinterpret_cast<float*>(b->h_src);
    launch_transform_cloud(d_points, n, m, d_xyz, b->stream);
    LOCGPU_HIP(ctx, hipGetLastError());
    // back in up to eight pieces, each with an event: the host writes piece i into the caller's points while the later ones cross PCIe —
    // the even pieces on the caller's thread, the odd ones on the context's helper thread (12 of every 32 bytes of a 3.7 MB array: the
    // write is what costs, and two cores do it in half the time)
    constexpr int kMaxPieces = 8;
    const size_t piece = std::max<size_t>(32 * 1024, (n + kMaxPieces - 1) / kMaxPieces);
    const int pieces = (int)((n + piece - 1) / piece);
    for (int p = 0; p < pieces; ++p) {
        if (!b->xyz_ev[p]) LOCGPU_HIP(ctx, hipEventCreateWithFlags(&b->xyz_ev[p], hipEventDisableTiming));
        const size_t lo = (size_t)p * piece, len = std::min(piece, n - lo);
        LOCGPU_HIP(ctx, hipMemcpyAsync(h_xyz + 3 * lo, d_xyz + 3 * lo, len * 3 * sizeof(float), hipMemcpyDeviceToHost, b->stream));
        LOCGPU_HIP(ctx, hipEventRecord(b->xyz_ev[p], b->stream));
    }
    char* ob = (char*)out;
    std::atomic<int> failed{0};
    auto scatter = [&, ob, h_xyz](int first) {
        for (int p = first; p < pieces; p += 2) {
            if (hipEventSynchronize(b->xyz_ev[p]) != hipSuccess) { failed = 1; return; }
            const size_t lo = (size_t)p * piece, hi = std::min(n, lo + piece);
            for (size_t i = lo; i < hi; ++i) std::memcpy(ob + i * out_stride, h_xyz + 3 * i, 12);
        }
    };
    if (ctx->worker) ctx->worker->wait();  // the copy of the other fields (scan_match_fields) is through
    const bool two = pieces > 1 && ctx->worker;
    if (two) ctx->worker->run([&] { (void)hipSetDevice(ctx->device); scatter(1); });
    scatter(0);
    if (two) ctx->worker->wait(); else if (pieces > 1) scatter(1);
    if (failed) return fail(ctx, LOCGPU_ERR_NO_DEVICE, "output cloud: copy back");
    return LOCGPU_OK;
}

// "*out = *src" of pcl::transformPointCloud — every field of a source point that the output point has room for — on the context's
// helper thread so that it runs beside the alignment; with out_fn the helper first ASKS for the output array (a façade sizes the
// caller's container there: 3.7 MB of first-touch page faults for a full scan, off the caller's thread). scan_match_output waits for
// it, then writes the coordinates. *dst receives the output array (written by the helper; read after wait()).
static void scan_match_fields(locgpu_ctx* ctx, const void* src, size_t n, size_t src_stride, void* out, size_t out_stride, locgpu_out_cloud_fn out_fn,
                              void* user, void** dst) {
    *dst = out;
    if (out_stride & LOCGPU_OUT_FIELDS_DONE) {  // the caller (or its callback) leaves the fields in place itself
        if (!out_fn) return;
        if (!ctx->worker) ctx->worker = new locgpu::HostWorker();
        ctx->worker->run([=] { *dst = out_fn(user, n); });
        return;
    }
    if ((!out && !out_fn) || (!out_fn && out == src) || n == 0) return;
    if (!ctx->worker) ctx->worker = new locgpu::HostWorker();
    ctx->worker->run([=] {
        void* o = out_fn ? out_fn(user, n) : out;
        *dst = o;
        if (!o || o == src) return;
        if (src_stride == out_stride) { std::memcpy(o, src, n * src_stride); return; }
        const size_t w = std::min(src_stride, out_stride);
        const char* sb = (const char*)src;
        char* ob = (char*)o;
        for (size_t i = 0; i < n; ++i) std::memcpy(ob + i * out_stride, sb + i * src_stride, w);
    });
}
static int scan_match_output(locgpu_ctx* ctx, locgpu_batch* b, size_t n, int rc, const double pose[7], void** dst, size_t out_stride) {
    if (ctx->worker) ctx->worker->wait();
    if (rc != LOCGPU_OK || !*dst) return rc;
    return write_output_cloud(ctx, b, b->d_src, n, pose, *dst, out_stride & ~LOCGPU_OUT_FIELDS_DONE);
}

int locgpu_icp_scan_match(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double init_pose[7], const locgpu_icp_opts* opts,
                          double out_pose[7], locgpu_align_stats* stats, void* out_cloud, size_t out_stride_bytes, locgpu_out_cloud_fn out_fn, void* out_user) {
    GnParams prm{};
    int k;
    float alpha_eff;
    int rc = check_icp(ctx, opts, prm, k, alpha_eff);
    if (rc != LOCGPU_OK) return rc;
    if (!src || !init_pose || !out_pose || ((out_cloud || out_fn) && (out_stride_bytes & ~LOCGPU_OUT_FIELDS_DONE) < 12)) return fail(ctx, LOCGPU_ERR_INVALID, "icp_scan_match: bad arguments");
    locgpu_batch* b = nullptr;
    rc = single_batch(ctx, src, n, stride_bytes, &b);
    if (rc != LOCGPU_OK) return rc;
    void* dst = nullptr;
    scan_match_fields(ctx, src, n, stride_bytes, out_cloud, out_stride_bytes, out_fn, out_user, &dst);
    rc = run_align(ctx, b, init_pose, prm, k, alpha_eff, false, out_pose, stats);
    return scan_match_output(ctx, b, n, rc, out_pose, &dst, out_stride_bytes);
}

int locgpu_ndt_scan_match(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double init_pose[7], double result_pose[7],
                          locgpu_align_stats* stats, void* out_cloud, size_t out_stride_bytes, locgpu_out_cloud_fn out_fn, void* out_user) {
    GnParams prm{};
    int rc = check_ndt(ctx, prm);
    if (rc != LOCGPU_OK) return rc;
    if (!src || !init_pose || !result_pose || ((out_cloud || out_fn) && (out_stride_bytes & ~LOCGPU_OUT_FIELDS_DONE) < 12)) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_scan_match: bad arguments");
    locgpu_batch* b = nullptr;
    rc = single_batch(ctx, src, n, stride_bytes, &b);
    if (rc != LOCGPU_OK) return rc;
    void* dst = nullptr;
    scan_match_fields(ctx, src, n, stride_bytes, out_cloud, out_stride_bytes, out_fn, out_user, &dst);
    double pose[7];
    locgpu_align_stats st{};
    rc = run_align(ctx, b, init_pose, prm, 0, 1.0f, true, pose, &st);
    if (stats) *stats = st;
    // det(H) == 0 (status 1): AlignNdt returns before it assigns result_pose (ndt_registration.cpp:435-436) — the caller's value stays,
    // and the output cloud is transformed by THAT pose (:258)
    if (rc == LOCGPU_OK && st.status != 1) std::memcpy(result_pose, pose, sizeof(pose));
    return scan_match_output(ctx, b, n, rc, result_pose, &dst, out_stride_bytes);
}

int locgpu_transform_cloud(locgpu_ctx* ctx, const double pose[7], const void* src, size_t n, size_t src_stride_bytes, void* out,
                           size_t out_stride_bytes) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!pose || (n && (!src || !out)) || src_stride_bytes < 12 || out_stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "transform_cloud: bad arguments");
    if (n == 0) return LOCGPU_OK;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    // through the context's one-scan batch: its pinned staging, its source array, its (idle) neighbour-list array — nothing is allocated per call
    locgpu_batch* b = nullptr;
    const int rc = single_batch(ctx, src, n, src_stride_bytes, &b);
    if (rc != LOCGPU_OK) return rc;
    return write_output_cloud(ctx, b, b->d_src, n, pose, out, out_stride_bytes);
}

// --------------------------------------------------------------------------------------------- measurement
int locgpu_graph_enable(locgpu_ctx* ctx, int on) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    ctx->use_graph = on != 0;
    return LOCGPU_OK;
}

int locgpu_profile_enable(locgpu_ctx* ctx, int on) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    ctx->profile = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return LOCGPU_OK;
}

int locgpu_profile_read(locgpu_ctx* ctx, double out[6], int reset) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    for (int j = 0; j < 3; ++j) {
        out[j] = ctx->prof_n[j] ? ctx->prof_ms[j] / (double)ctx->prof_n[j] : 0.0;
        out[3 + j] = (double)ctx->prof_n[j];
    }
    if (reset)
        for (int j = 0; j < 3; ++j) { ctx->prof_ms[j] = 0; ctx->prof_n[j] = 0; }
    return LOCGPU_OK;
}

int locgpu_search_stats_read(locgpu_ctx* ctx, uint64_t out[4], int reset) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_search_stats) {
        LOCGPU_HIP(ctx, hipMalloc((void**)&ctx->d_search_stats, kSearchStatSlots * sizeof(unsigned long long)));
        if (!fill_now(ctx, ctx->d_search_stats, kSearchStatSlots * sizeof(unsigned long long), "hipMemset search stats")) return LOCGPU_ERR_NO_DEVICE;
    }
    for (hipStream_t st : ctx->slot_stream) LOCGPU_HIP(ctx, hipStreamSynchronize(st));
    unsigned long long h[kSearchStatSlots];
    LOCGPU_HIP(ctx, hipMemcpy(h, ctx->d_search_stats, sizeof(h), hipMemcpyDeviceToHost));
    static const bool stamp = getenv("LOCGPU_STAMP") != nullptr;
    // diagnostic build (LOCGPU_STAMP=1): out[2] = Σ over queries of the main-loop rounds the query needed, out[3] = Σ over queries of
    // the rounds its wave ran (what the wave paid for that lane) — lane efficiency of the search kernel = out[2] / out[3]
    out[0] = h[0]; out[1] = h[1]; out[2] = stamp ? h[4] : h[2]; out[3] = stamp ? h[13] : 0;
    if (stamp && h[12])
        fprintf(stderr, "[locgpu stamp] %llu queries in %llu waves: %.2f rounds per lane, %.2f per wave; lane efficiency %.3f\n", h[0], h[12], (double)h[4] / (double)h[0],
                (double)h[9] / (double)h[12], (double)h[4] / (double)std::max<unsigned long long>(h[13], 1ull));
    if (reset && !fill_now(ctx, ctx->d_search_stats, sizeof(h), "hipMemset search stats")) return LOCGPU_ERR_NO_DEVICE;
    return LOCGPU_OK;
}

int locgpu_visit_count_enable(locgpu_ctx* ctx, int on) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->d_visits) {
        LOCGPU_HIP(ctx, hipMalloc((void**)&ctx->d_visits, 4 * sizeof(unsigned long long)));
        if (!fill_now(ctx, ctx->d_visits, 4 * sizeof(unsigned long long), "hipMemset visits")) return LOCGPU_ERR_NO_DEVICE;
    }
    ctx->count_visits = on != 0;
    return LOCGPU_OK;
}

int locgpu_visit_count_read(locgpu_ctx* ctx, uint64_t out[4], int reset) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (!ctx->d_visits) return LOCGPU_OK;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    for (hipStream_t st : ctx->slot_stream) LOCGPU_HIP(ctx, hipStreamSynchronize(st));
    unsigned long long h[4];
    LOCGPU_HIP(ctx, hipMemcpy(h, ctx->d_visits, sizeof(h), hipMemcpyDeviceToHost));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = h[3];
    if (reset && !fill_now(ctx, ctx->d_visits, sizeof(h), "hipMemset visits")) return LOCGPU_ERR_NO_DEVICE;
    return LOCGPU_OK;
}

}  // extern "C"

// Diagnostic build only (LOCGPU_STAMP=1 when the batch was created): per query the main-loop rounds it needed in the batch's most
// recent search stage, out[n_scans * max_n].
extern "C" __attribute__((visibility("default"))) int locgpu_debug_stamp_trips(locgpu_ctx* ctx, locgpu_batch* b, uint32_t* out) {
    if (!ctx || !b || !out || !getenv("LOCGPU_STAMP")) return LOCGPU_ERR_INVALID;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    LOCGPU_HIP(ctx, hipStreamSynchronize(b->stream));
    LOCGPU_HIP(ctx, hipMemcpy(out, b->d_redo_list + b->pitch, b->pitch * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return LOCGPU_OK;
}

// Test hook (not part of include/locgpu.h): the device-built exact-search grid of the current target, copied out so the test-suite
// can check its invariants. info = {dims x,y,z, occupied cells, occupied tiles, hash capacity, tile dims x,y,z};
// params = {origin x,y,z, cell, inv_cell, slack}; pts: n × 4 floats (x, y, z, bits(leaf slot)) in (tile, cell) order; tiles: the raw
// 140-byte TileRec records; hash: {tile_lin, record} × capacity. Returns the number of grid points (0 on failure).
extern "C" __attribute__((visibility("default"))) size_t locgpu_debug_grid_dump(locgpu_ctx* ctx, int64_t info[9], float params[6], float* pts, size_t pts_cap,
                                                                                 void* tiles, size_t tiles_cap_bytes, uint32_t* hash, size_t hash_cap) {
    if (!ctx || target_join(ctx) != LOCGPU_OK || !ctx->d_tree || ensure_grid(ctx) != LOCGPU_OK) return 0;
    const locgpu::GridView& g = ctx->grid;
    if (info) {
        for (int a = 0; a < 3; ++a) { info[a] = g.dims[a]; info[6 + a] = g.tdims[a]; }
        info[3] = (int64_t)g.num_cells; info[4] = g.n_tocc; info[5] = (int64_t)g.tile_mask + 1;
    }
    if (params) { for (int a = 0; a < 3; ++a) params[a] = g.origin[a]; params[3] = g.cell; params[4] = g.inv_cell; params[5] = g.slack; }
    if (pts && hipMemcpy(pts, g.pts, std::min(pts_cap, g.num_points * 4) * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (tiles && hipMemcpy(tiles, g.tiles, std::min(tiles_cap_bytes, (size_t)g.n_tocc * sizeof(locgpu::TileRec)), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    if (hash && hipMemcpy(hash, g.tile_hash, std::min(hash_cap, ((size_t)g.tile_mask + 1) * 2) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return g.num_points;
}

// --------------------------------------------------------------------------------------------- NDT
namespace locgpu {
void ndt_free(locgpu_ctx* ctx) {
    if (ctx->ndt) { ndt_table_free(*ctx->ndt); delete ctx->ndt; ctx->ndt = nullptr; }
    if (ctx->inc) { inc_ndt_destroy(ctx->inc); ctx->inc = nullptr; }
}
int check_ndt(locgpu_ctx* ctx, GnParams& prm) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    const bool inc = ctx->ndt_opts.method == 2;
    if ((inc && !ctx->inc) || (!inc && !ctx->ndt)) return fail(ctx, LOCGPU_ERR_NO_TARGET, "ndt: SetInputTarget has not been called");
    prm.method = inc ? 4 : 3;
    prm.max_iteration = ctx->ndt_opts.max_iteration;
    prm.min_effective_pts = ctx->ndt_opts.min_effective_pts;
    prm.eps = ctx->ndt_opts.eps;
    prm.max_nn_distance = prm.max_plane_distance = prm.max_line_distance = 0.0;
    return LOCGPU_OK;
}
}  // namespace locgpu

extern "C" {

// Target ingest from a cloud already in HBM. `host` (optional) = the same points on the host: the incremental method replays a
// cloud whose own working set exceeds the voxel capacity point by point there (ndt_inc.hip) and fetches them when absent.
static int ndt_set_target_dev(locgpu_ctx* ctx, const float4* d_pts, const float4* host, size_t n, const locgpu_ndt_opts* opts) {
    locgpu_ndt_opts o;
    if (opts) o = *opts; else locgpu_ndt_opts_default(&o);
    if (!(o.voxel_size > 0.0) || (o.nearby_type != 0 && o.nearby_type != 1) || (o.method != 1 && o.method != 2) || (o.method == 2 && o.capacity < 2))
        return fail(ctx, LOCGPU_ERR_INVALID, "ndt_set_target: bad options");
    if (o.method == 2) {
        // incremental: keep the voxel set unless the grid itself changed
        if (ctx->inc && (ctx->ndt_opts.method != 2 || ctx->ndt_opts.voxel_size != o.voxel_size || ctx->ndt_opts.capacity != o.capacity)) {
            inc_ndt_destroy(ctx->inc);
            ctx->inc = nullptr;
        }
        if (!ctx->inc) ctx->inc = inc_ndt_create((size_t)o.capacity, o.voxel_size);
        bool bad = false;
        const hipError_t e = inc_ndt_ingest(*ctx->inc, host, d_pts, n, ctx->stream, &bad);
        if (e != hipSuccess) {
            // an ingest that failed half way (its counters, slot arrays, free stack and table may disagree): the voxel set is gone,
            // the next SetInputTarget starts from an empty one
            (void)hipStreamSynchronize(ctx->stream);
            inc_ndt_destroy(ctx->inc);
            ctx->inc = nullptr;
            hip_ok(ctx, e, "inc_ndt_ingest");
            return LOCGPU_ERR_NO_DEVICE;
        }
        ctx->ndt_opts = o;
        ctx->target_epoch++;
        if (bad) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_set_target: a point lies outside the +-2^20-voxel key range (it was skipped)");
        return LOCGPU_OK;
    }
    if (ctx->inc) { inc_ndt_destroy(ctx->inc); ctx->inc = nullptr; }
    if (!ctx->ndt) ctx->ndt = new NdtTable();
    bool bad_key = false;
    const hipError_t e = ndt_build(*ctx->ndt, d_pts, n, o.voxel_size, o.min_pts_in_voxel, ctx->stream, &bad_key);
    if (e != hipSuccess) { ndt_free(ctx); hip_ok(ctx, e, "ndt_build"); return LOCGPU_ERR_NO_DEVICE; }
    if (bad_key) { ndt_free(ctx); return fail(ctx, LOCGPU_ERR_INVALID, "ndt_set_target: a point lies outside the +-2^20-voxel key range"); }
    ctx->ndt->res_outlier_th = o.res_outlier_th;
    ctx->ndt->n_nearby = o.nearby_type == 0 ? 1 : 7;
    ctx->ndt_opts = o;
    ctx->target_epoch++;
    return LOCGPU_OK;
}

int locgpu_ndt_set_target(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, const locgpu_ndt_opts* opts) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!pts || n == 0 || stride_bytes < 12) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_set_target: empty cloud or stride < 12");
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<float4> host(n);
    const char* base = (const char*)pts;
    for (size_t i = 0; i < n; ++i) { host[i] = float4{0.f, 0.f, 0.f, 0.f}; std::memcpy(&host[i], base + i * stride_bytes, 12); }
    float4* d_pts = nullptr;
    LOCGPU_HIP(ctx, hipMalloc((void**)&d_pts, n * sizeof(float4)));
    if (!hip_ok(ctx, hipMemcpy(d_pts, host.data(), n * sizeof(float4), hipMemcpyHostToDevice), "H2D map")) { (void)hipFree(d_pts); return LOCGPU_ERR_NO_DEVICE; }
    const int rc = ndt_set_target_dev(ctx, d_pts, host.data(), n, opts);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_pts);
    return rc;
}

int locgpu_ndt_target_info(const locgpu_ctx* ctx, int64_t out[3]) {
    if (!ctx || !out) return LOCGPU_ERR_INVALID;
    if (ctx->ndt_opts.method == 2 && ctx->inc) { out[0] = (int64_t)inc_ndt_num_voxels(ctx->inc); out[1] = ctx->ndt_opts.capacity; out[2] = 0; return LOCGPU_OK; }
    if (!ctx->ndt) { out[0] = out[1] = out[2] = 0; return LOCGPU_ERR_NO_TARGET; }
    out[0] = (int64_t)ctx->ndt->n_vox;
    out[1] = (int64_t)ctx->ndt->cap;
    out[2] = (int64_t)(ctx->ndt->cap * sizeof(NdtSlot) + ctx->ndt->n_vox * sizeof(NdtRecord));
    return LOCGPU_OK;
}

int locgpu_ndt_dump(locgpu_ctx* ctx, int32_t* keys, double* mu, double* info, size_t cap, size_t* n_out) {
    if (!ctx || !n_out) return LOCGPU_ERR_INVALID;
    if (ctx->ndt_opts.method == 2 && ctx->inc) {
        LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
        LOCGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *n_out = inc_ndt_dump(ctx->inc, keys, mu, info, cap);
        return LOCGPU_OK;
    }
    if (!ctx->ndt) return fail(ctx, LOCGPU_ERR_NO_TARGET, "ndt_dump: no target");
    *n_out = ctx->ndt->n_vox;
    const size_t n = std::min(cap, ctx->ndt->n_vox);
    if (n == 0) return LOCGPU_OK;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    LOCGPU_HIP(ctx, ndt_dump(*ctx->ndt, keys, mu, info, n, ctx->stream));
    return LOCGPU_OK;
}

int locgpu_ndt_align_batch(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses, double* out_poses, locgpu_align_stats* stats) {
    GnParams prm{};
    const int rc = check_ndt(ctx, prm);
    if (rc != LOCGPU_OK) return rc;
    if (!b || b->ctx != ctx || !init_poses || !out_poses) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_align_batch: bad arguments");
    return run_align(ctx, b, init_poses, prm, 0, 1.0f, true, out_poses, stats);
}

int locgpu_ndt_align_batch_begin(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses) {
    GnParams prm{};
    const int rc = check_ndt(ctx, prm);
    if (rc != LOCGPU_OK) return rc;
    if (!b || b->ctx != ctx || !init_poses) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_align_batch_begin: bad arguments");
    return align_begin(ctx, b, init_poses, prm, 0, 1.0f, true);
}

int locgpu_ndt_align(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double init_pose[7], double out_pose[7],
                     locgpu_align_stats* stats) {
    GnParams prm{};
    int rc = check_ndt(ctx, prm);
    if (rc != LOCGPU_OK) return rc;
    if (!src || !init_pose || !out_pose) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_align: bad arguments");
    locgpu_batch* b = nullptr;
    rc = single_batch(ctx, src, n, stride_bytes, &b);
    if (rc != LOCGPU_OK) return rc;
    return run_align(ctx, b, init_pose, prm, 0, 1.0f, true, out_pose, stats);
}

// ---- matcher entry points on clouds resident in HBM (cloud_filters.hpp) ----
int locgpu_icp_set_target_cloud(locgpu_ctx* ctx, const locgpu_cloud* target) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!target || target->ctx != ctx) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target_cloud: bad cloud");
    if (target->n == 0) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target: empty cloud or stride < 12");
    (void)target_join(ctx, false);  // a pending asynchronous ingest is superseded
    // The mean-split tree is built on the host (its float32 sums are sequential by definition, kdtree.cpp:94-123), so the
    // cloud crosses PCIe once in each direction: 16 B/point down, the packed tree (≈24 B/point) up.
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    float4* stage = nullptr;
    if (!hip_ok(ctx, cloud_stage(ctx, target->n, &stage), "pinned staging")) return LOCGPU_ERR_OOM;
    LOCGPU_HIP(ctx, hipMemcpyAsync(stage, target->d, target->n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    LOCGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return locgpu_icp_set_target(ctx, stage, target->n, sizeof(float4));
}

int locgpu_icp_set_target_cloud_async(locgpu_ctx* ctx, const locgpu_cloud* target) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!target || target->ctx != ctx) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target_cloud: bad cloud");
    if (target->n == 0) return fail(ctx, LOCGPU_ERR_INVALID, "icp_set_target: empty cloud or stride < 12");
    (void)target_join(ctx, false);  // an earlier pending ingest is superseded
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    float4* stage = nullptr;
    if (!hip_ok(ctx, cloud_stage(ctx, target->n, &stage), "pinned staging")) return LOCGPU_ERR_OOM;
    LOCGPU_HIP(ctx, hipMemcpyAsync(stage, target->d, target->n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    LOCGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    locgpu::PendingTarget* p = take_target_scratch(ctx);
    p->xyz.resize(3 * target->n);  // the deep copy of SetInputTarget (icp_registration.cpp:16): the staging block is free again after it
    for (size_t i = 0; i < target->n; ++i) std::memcpy(&p->xyz[3 * i], &stage[i], 12);
    const size_t n = target->n;
    p->worker = std::thread([p, n] { p->ok = build_packed_kdtree(p->xyz.data(), n, p->tree, p->err); });
    ctx->pending_target = p;
    return LOCGPU_OK;
}

int locgpu_ndt_set_target_cloud(locgpu_ctx* ctx, const locgpu_cloud* target, const locgpu_ndt_opts* opts) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!target || target->ctx != ctx) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_set_target_cloud: bad cloud");
    if (target->n == 0) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_set_target: empty cloud or stride < 12");
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    const int rc = ndt_set_target_dev(ctx, target->d, nullptr, target->n, opts);
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

int locgpu_icp_align_cloud(locgpu_ctx* ctx, const locgpu_cloud* src, const double init_pose[7], const locgpu_icp_opts* opts, double out_pose[7],
                           locgpu_align_stats* stats) {
    GnParams prm{};
    int k;
    float alpha_eff;
    int rc = check_icp(ctx, opts, prm, k, alpha_eff);
    if (rc != LOCGPU_OK) return rc;
    if (!src || !src->ctx || !init_pose || !out_pose) return fail(ctx, LOCGPU_ERR_INVALID, "icp_align_cloud: bad arguments");
    { const hipError_t ce = cloud_input_ready(ctx, src); if (ce == hipErrorInvalidDevice) return fail(ctx, LOCGPU_ERR_INVALID, "icp_align_cloud: the cloud belongs to a context on another GPU"); if (!hip_ok(ctx, ce, "icp_align_cloud: ordering behind the cloud's context")) return LOCGPU_ERR_NO_DEVICE; }
    locgpu_batch* b = nullptr;
    rc = single_batch_dev(ctx, src->d, src->n, &b);
    if (rc != LOCGPU_OK) return rc;
    return run_align(ctx, b, init_pose, prm, k, alpha_eff, false, out_pose, stats);
}

int locgpu_ndt_align_cloud(locgpu_ctx* ctx, const locgpu_cloud* src, const double init_pose[7], double out_pose[7], locgpu_align_stats* stats) {
    GnParams prm{};
    int rc = check_ndt(ctx, prm);
    if (rc != LOCGPU_OK) return rc;
    if (!src || !src->ctx || !init_pose || !out_pose) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_align_cloud: bad arguments");
    { const hipError_t ce = cloud_input_ready(ctx, src); if (ce == hipErrorInvalidDevice) return fail(ctx, LOCGPU_ERR_INVALID, "ndt_align_cloud: the cloud belongs to a context on another GPU"); if (!hip_ok(ctx, ce, "ndt_align_cloud: ordering behind the cloud's context")) return LOCGPU_ERR_NO_DEVICE; }
    locgpu_batch* b = nullptr;
    rc = single_batch_dev(ctx, src->d, src->n, &b);
    if (rc != LOCGPU_OK) return rc;
    return run_align(ctx, b, init_pose, prm, 0, 1.0f, true, out_pose, stats);
}

}  // extern "C"

