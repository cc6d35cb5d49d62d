/* include/locgpu.h — C ABI of the MI355X-native registration hot path (liblocgpu.so).
 *
 * This is the drop-in boundary for the ONE path of maotian123/loc_lib this project accelerates:
 * the KD-tree neighbour search + point-to-plane/-line/-point ICP and direct-NDT Gauss–Newton loop
 * that slam_demo's front-ends run once per scan through LocUtils::MatchingInterface. The reference
 * is plain C++ with no FFI; a maintainer binds these entry points from the reference's own matcher
 * classes (see INTEGRATION.md and loc_lib_amd/host/). Every entry point names the reference interface it
 * replaces (paths relative to the reference repo root).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch/Eigen/PCL types cross this boundary;
 *   - clouds are passed as (base pointer, point count, stride in BYTES); x,y,z are three consecutive
 *     float32 at the start of each point (pcl::PointXYZI: stride 32; packed xyz: stride 12);
 *   - poses are 7 doubles, quaternion (x,y,z,w) then translation (x,y,z): the memory layout of
 *     Sophus::SE3d::data() (LocUtils/include/LocUtils/common/eigen_types.h:66, `using SE3 = Sophus::SE3d`);
 *   - host-pointer entry points copy their inputs before returning (the reference deep-copies target
 *     and source: icp_registration.cpp:16,259), are synchronous, and are not re-entrant per context
 *     (the reference matcher is single-threaded per instance);
 *   - every function returns LOCGPU_OK (0) or a negative locgpu_status; no exception crosses the boundary.
 *     locgpu_last_error() gives the text. There is NO CPU fallback: without a HIP device every call fails.
 */
#ifndef LOCGPU_H_
#define LOCGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LOCGPU_API __attribute__((visibility("default")))

typedef struct locgpu_ctx locgpu_ctx;     /* one matcher instance (tree / voxel grid + workspaces) on one GPU */
typedef struct locgpu_batch locgpu_batch; /* a batch of scans resident in HBM */

typedef enum locgpu_status {
    LOCGPU_OK = 0,
    LOCGPU_ERR_INVALID = -1,    /* bad argument */
    LOCGPU_ERR_NO_DEVICE = -2,  /* no HIP device / HIP runtime error */
    LOCGPU_ERR_NO_TARGET = -3,  /* align/search before set_target */
    LOCGPU_ERR_K_TOO_LARGE = -4,/* k > number of tree leaves (kdtree.cpp:149-153 logs an error and returns false) */
    LOCGPU_ERR_DEPTH = -5,      /* tree deeper than the traversal stack supports */
    LOCGPU_ERR_OOM = -6
} locgpu_status;

/* IcpMethod, LocUtils/include/LocUtils/model/matching/3d/icp/icp_registration.hpp:15-20 (PCLICP is not on the path). */
typedef enum locgpu_icp_method { LOCGPU_P2P = 0, LOCGPU_P2LINE = 1, LOCGPU_P2PLANE = 2 } locgpu_icp_method;

/* How correspondences are searched.
 * TREE_FAITHFUL replays the reference's mean-split KD-tree and its DFS visit order bit for bit
 * (kdtree.cpp:169-236), including the alpha-pruned approximate mode that is the reference default
 * (kdtree.h:128-129). GRID_EXACT is the exact k-NN over a radix-sorted voxel grid: it equals the
 * reference with KdtreeRegistration::SetEnableANN(false) (kdtree.cpp:285-288) up to ties in distance. */
typedef enum locgpu_search_mode { LOCGPU_SEARCH_TREE_FAITHFUL = 0, LOCGPU_SEARCH_GRID_EXACT = 1 } locgpu_search_mode;

/* IcpOptions, icp_registration.hpp:22-39 (same defaults via locgpu_icp_opts_default). */
typedef struct locgpu_icp_opts {
    int32_t method;            /* locgpu_icp_method; reference default P2P */
    int32_t max_iteration;     /* 20 */
    double max_nn_distance;    /* 1.0  (compared with a SQUARED distance, icp cpp:75) */
    double max_plane_distance; /* 0.1 */
    double max_line_distance;  /* 0.5 */
    int32_t min_effective_pts; /* 10 */
    double eps;                /* 1e-2 */
    int32_t approximate;       /* 1: KdTree::approximate_ (kdtree.h:128) */
    float ann_alpha;           /* 0.1f: KdTree::alpha_ (kdtree.h:129) */
    int32_t search_mode;       /* locgpu_search_mode */
} locgpu_icp_opts;

/* NdtOptions, LocUtils/include/LocUtils/model/matching/3d/ndt/ndt_registration.hpp:27-42. */
typedef struct locgpu_ndt_opts {
    int32_t max_iteration;     /* 20 */
    double voxel_size;         /* 1.0; inv_voxel_size_ is always recomputed as 1/voxel_size (ndt cpp:15,25) */
    int32_t min_effective_pts; /* 10 */
    int32_t min_pts_in_voxel;  /* 3 (a voxel is kept iff count > this) */
    double eps;                /* 1e-2 */
    double res_outlier_th;     /* 20.0 */
    int32_t nearby_type;       /* 0 CENTER, 1 NEARBY6 (hpp:16-20) */
    int32_t method;            /* 1 DIRECT_NDT (default), 2 INCREMENTAL_NDT (NdtMethod, hpp:21-26; 0 PCL_NDT is a no-op in the reference) */
    int64_t capacity;          /* 100000: LRU voxel capacity of the incremental variant (capacity_, hpp:37) */
} locgpu_ndt_opts;

/* Per-scan result counters (the reference only logs these through glog). */
typedef struct locgpu_align_stats {
    int32_t iterations;        /* Gauss–Newton iterations executed (H,B evaluations) */
    int32_t converged;         /* 1 if the loop left through |dx| < eps */
    int32_t status;            /* 0 ok; 1 direct-NDT det(H)==0 ⇒ reference returns before writing result_pose (ndt cpp:435-436);
                                  2 incremental NDT: too few effective residuals ⇒ returns false with the current pose (ndt cpp:349-353) */
    int32_t reserved;
    int64_t last_effective_num;
    double last_dx_norm;
} locgpu_align_stats;

LOCGPU_API void locgpu_icp_opts_default(locgpu_icp_opts* o);
LOCGPU_API void locgpu_ndt_opts_default(locgpu_ndt_opts* o);

/* Lifetime. A context is what one IcpRegistration / NdtRegistration instance owns (icp_registration.hpp:41-142). */
LOCGPU_API int locgpu_create(int device_id, locgpu_ctx** out);
LOCGPU_API void locgpu_destroy(locgpu_ctx* ctx);
LOCGPU_API const char* locgpu_last_error(const locgpu_ctx* ctx); /* ctx may be NULL: error of the last failed create */
LOCGPU_API int locgpu_device_count(void);

/* ---- ICP target: IcpRegistration::SetInputTarget (icp_registration.cpp:9-29) →
 *      KdtreeRegistration::SetTargetCloud / KdTree::BuildTree (kdtree.cpp:261-270, 10-31). Host pointer. */
LOCGPU_API int locgpu_icp_set_target(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes);
/* The same with the host tree build on a worker thread: returns once the points have been copied (the deep copy the reference makes,
 * icp_registration.cpp:16); the first entry point that reads the ICP target completes the ingest on the caller's thread and reports
 * its errors. See locgpu_icp_set_target_cloud_async. */
LOCGPU_API int locgpu_icp_set_target_async(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes);
/* out[0]=leaves (KdTree::size_), out[1]=tree nodes, out[2]=depth, out[3]=bytes of the packed tree in HBM */
LOCGPU_API int locgpu_icp_target_info(const locgpu_ctx* ctx, int64_t out[4]);

/* ---- SearchPointInterface::FindNearstPoints (search_point_interface.h:13; kdtree.cpp:272-283), many queries at once.
 * queries: nq × 3 packed float32 (host). out_idx: nq × k int32 original point indices, ascending distance (host).
 * visits (optional, host): nq × 2 uint32 {tree nodes visited, leaves visited} per query. */
LOCGPU_API int locgpu_knn(locgpu_ctx* ctx, const float* queries, size_t nq, int k, int approximate, float alpha, int search_mode,
                          int32_t* out_idx, uint32_t* visits);

/* ---- BfnnRegistration (LocUtils/include/LocUtils/model/search_point/bfnn/bfnn.h:11-37, bfnn.cpp:14-50): the brute-force
 * implementation of SearchPointInterface. set_target = SetTargetCloud (deep copy; false/-1 for an empty cloud); knn = FindNearstPoints
 * for nq queries at once: out_idx nq × k original point indices in ascending float32 distance. Equal distances — an order the
 * reference's std::sort leaves open — are ordered by index. k > cloud size is an error here (the reference reads past the end). */
LOCGPU_API int locgpu_bfnn_set_target(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes);
LOCGPU_API int locgpu_bfnn_knn(locgpu_ctx* ctx, const float* queries, size_t nq, int k, int32_t* out_idx);

/* ---- MatchingInterface::CaculateMatrixHAndB (matching_interface.h:18-24; icp_registration.cpp:31-55): one
 * evaluation of H (6×6 row-major) and B at `pose`. *ok receives the reference's bool (false: too few effective
 * points or det(H)==0). Used by LoamRegistration (loam_registration.cpp:56,66). */
LOCGPU_API int locgpu_icp_hb(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double pose[7],
                             const locgpu_icp_opts* opts, double H[36], double B[6], int64_t* effective_num, int* ok);

/* ---- IcpRegistration::ScanMatch minus the output cloud (icp_registration.cpp:216-239 → AlignP2P/P2Line/P2Plane :267-381). */
LOCGPU_API int locgpu_icp_align(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double init_pose[7],
                                const locgpu_icp_opts* opts, double out_pose[7], locgpu_align_stats* stats);

/* ---- IcpRegistration::ScanMatch WHOLE (icp_registration.cpp:216-244): the alignment and the output cloud
 * `pcl::transformPointCloud(*input_source, *result_cloud_ptr, result_pose.matrix().cast<float>())` (:241) in one call — the source is
 * uploaded once, the transform runs on the copy that is in HBM from the alignment, and x, y, z come back through pinned staging.
 * The output cloud: n points of out_stride_bytes each, given either as out_cloud or by out_fn — a callback the library calls ONCE, on a
 * helper thread, while the alignment runs, with out_user and the point count; it returns the output array (the façade sizes the caller's
 * pcl::PointCloud there: the first touch of a fresh 3.7 MB array is then off the caller's thread) or NULL for none. Both NULL: no
 * output cloud. Like pcl::transformPointCloud every output point is the source point with x, y, z replaced: unless the output array
 * IS the source array (same stride: in place) the first min(stride_bytes, out_stride_bytes) bytes of every source point are copied
 * — on the helper thread, beside the alignment. Any other overlap of the two clouds is undefined. out_stride_bytes | LOCGPU_OUT_FIELDS_DONE:
 * the output points already hold their other fields (or the callback puts them there — the façade's does `out->points = src->points`,
 * one pass over fresh memory instead of a resize and a copy): the library then only writes x, y, z. The callback must not call into
 * this context. This is what LocUtils::IcpRegistration::ScanMatch binds to. */
typedef void* (*locgpu_out_cloud_fn)(void* user, size_t n_points);
#define LOCGPU_OUT_FIELDS_DONE ((size_t)1 << (sizeof(size_t) * 8 - 1))
LOCGPU_API int locgpu_icp_scan_match(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double init_pose[7],
                                     const locgpu_icp_opts* opts, double out_pose[7], locgpu_align_stats* stats, void* out_cloud,
                                     size_t out_stride_bytes, locgpu_out_cloud_fn out_fn, void* out_user);
/* ---- NdtRegistration::ScanMatch WHOLE (ndt_registration.cpp:238-261). result_pose is IN-OUT like the reference's `SE3& result_pose`:
 * with stats->status == 1 (det(H) == 0) AlignNdt returns before it assigns it (:435-436), so the caller's value stays and the output
 * cloud is transformed by that value (:258); otherwise it receives the result. Output cloud as above. */
LOCGPU_API int locgpu_ndt_scan_match(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double init_pose[7],
                                     double result_pose[7], locgpu_align_stats* stats, void* out_cloud, size_t out_stride_bytes,
                                     locgpu_out_cloud_fn out_fn, void* out_user);

/* ---- pcl::transformPointCloud(*src, *out, pose.matrix().cast<float>()) (icp_registration.cpp:241, ndt_registration.cpp:258).
 * Writes x,y,z of each output point (float32 arithmetic); other fields of the output points are left untouched. */
LOCGPU_API int locgpu_transform_cloud(locgpu_ctx* ctx, const double pose[7], const void* src, size_t n, size_t src_stride_bytes,
                                      void* out, size_t out_stride_bytes);

/* ---- Batched many-scans-vs-one-map mode (BASELINE.json config 4; no reference counterpart — the reference loops
 * ScanMatch over scans). Scans are uploaded once and stay resident; each align call runs every scan's own GN loop. */
LOCGPU_API int locgpu_batch_create(locgpu_ctx* ctx, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n_scans,
                                   locgpu_batch** out);
LOCGPU_API void locgpu_batch_destroy(locgpu_batch* b);
/* The source deep copy of ScanMatch (SetSource, icp_registration.cpp:221,252-265) for a whole batch, overlapped with the GPU's work:
 * locgpu_batch_create_empty reserves room for n_scans scans of at most max_points_per_scan points; locgpu_batch_upload_async
 * replaces the batch's scans (counts[s] <= max_points_per_scan) and returns at once — a worker packs the strided host points
 * into pinned slots and streams them to HBM on the context's copy stream, under whatever its compute streams
 * are running (e.g. the align call of ANOTHER batch: two batches alternate as a double buffer; one upload per context at a
 * time — a second one waits for the first one's packing). The host clouds must stay valid
 * until locgpu_batch_upload_wait returns (it returns the upload's status); every align / hb call on the batch waits for its
 * pending upload first. An upload into a batch whose alignment has been begun and not ended (locgpu_*_align_batch_begin) is
 * refused with LOCGPU_ERR_INVALID — rotate one more batch than there are alignments in flight. A failed upload stays with its
 * batch: that batch's next align / hb / upload_wait returns the failure until a new upload replaces its scans. */
LOCGPU_API int locgpu_batch_create_empty(locgpu_ctx* ctx, int n_scans, size_t max_points_per_scan, locgpu_batch** out);
LOCGPU_API int locgpu_batch_upload_async(locgpu_batch* b, const void* const* srcs, const size_t* counts, size_t stride_bytes);
LOCGPU_API int locgpu_batch_upload_wait(locgpu_batch* b);
/* init_poses / out_poses: n_scans × 7 doubles (host). stats: n_scans entries or NULL. */
LOCGPU_API int locgpu_icp_align_batch(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses, const locgpu_icp_opts* opts,
                                      double* out_poses, locgpu_align_stats* stats);
LOCGPU_API int locgpu_ndt_align_batch(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses, double* out_poses,
                                      locgpu_align_stats* stats);
/* The same alignments in two halves, for a caller that keeps several batches in flight — up to three pay — (no reference counterpart: the reference
 * matches one scan at a time). The batches of a context are dealt to three compute streams in turn; *_begin copies the poses,
 * enqueues the first eight Gauss–Newton iterations on the batch's stream and returns, locgpu_align_batch_end waits for them,
 * enqueues further iterations while scans are still open and writes the results. Begun on batch B while batch A is not yet
 * ended, B's first iterations fill the chip under A's last ones (which hold a handful of unconverged scans):
 *     begin(A); loop { begin(B); end(A); swap(A, B); }
 * One alignment per batch at a time and no locgpu_batch_upload_async into the batch between begin and end (LOCGPU_ERR_INVALID);
 * results are those of the blocking calls, bit for bit. The target must not change between begin and end. Sharded batches: every rank begins and ends its batches in the same order. */
LOCGPU_API int locgpu_icp_align_batch_begin(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses, const locgpu_icp_opts* opts);
LOCGPU_API int locgpu_ndt_align_batch_begin(locgpu_ctx* ctx, locgpu_batch* b, const double* init_poses);
LOCGPU_API int locgpu_align_batch_end(locgpu_ctx* ctx, locgpu_batch* b, double* out_poses, locgpu_align_stats* stats);
/* One H,B evaluation for every scan of the batch at the given poses (point-sharded multi-GPU mode: the caller
 * all-reduces hb over ranks, then calls locgpu_gn_update). hb: n_scans × 44 doubles = H36, B6, effective_num, ok. */
LOCGPU_API int locgpu_icp_hb_batch(locgpu_ctx* ctx, locgpu_batch* b, const double* poses, const locgpu_icp_opts* opts, double* hb);
/* The update step of AlignP2Plane (icp_registration.cpp:362-375) on already-reduced normal equations:
 * returns 1 in *stop when |dx| < eps. method selects the P2P "/16" quirk (icp cpp:287). */
LOCGPU_API int locgpu_gn_update(const double hb[44], int method, int min_effective_pts, double eps, double pose[7], double dx[6],
                                int* applied, int* stop);

/* ---- One node, several GPUs (BASELINE.json configs[3]; no reference counterpart): one process per GPU, one context per process,
 * the ranks joined by an RCCL communicator. A SHARDED batch has n_total scans; this rank holds the points of scans
 * [first_scan, first_scan + n_local) — or, for one large alignment split by points, a slice of the points of every scan
 * (first_scan = 0, n_local = n_total). Poses, convergence flags and normal equations exist for all n_total scans on every rank:
 * in each Gauss–Newton iteration the per-scan sums (21 H + 6 B + effective_num; zeros for scans a rank does not hold) are
 * all-reduced over xGMI (on the context's communication stream, in host order) and every rank solves every scan, so all ranks take the same decisions.
 * The align / hb entry points are then COLLECTIVE (every rank calls them with the same poses and options; init_poses, out_poses
 * and stats have n_total entries) and scan-sharded results are bit-identical to the single-GPU ones. */
#define LOCGPU_COMM_ID_BYTES 128
LOCGPU_API int locgpu_comm_unique_id(void* id_out /* LOCGPU_COMM_ID_BYTES, made on one rank and handed to all */);
LOCGPU_API int locgpu_comm_init(locgpu_ctx* ctx, int rank, int world, const void* id);
LOCGPU_API int locgpu_comm_info(const locgpu_ctx* ctx, int* rank, int* world);
LOCGPU_API int locgpu_batch_create_sharded(locgpu_ctx* ctx, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n_local,
                                           int first_scan, int n_total, locgpu_batch** out);
/* IcpRegistration::SetInputTarget, collective: rank `root` builds the KD-tree from ITS pts (the others' are ignored) and
 * broadcasts the packed tree — one host build per node instead of one per GPU. */
LOCGPU_API int locgpu_icp_set_target_bcast(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, int root);

/* ---- Open-scan pool (round 5; no reference counterpart): the batched many-scans-vs-one-map mode as a continuous service.
 * The reference's loops stop per scan (icp_registration.cpp:358-376, ndt_registration.cpp:393-462): the late Gauss–Newton
 * iterations of a batch hold a handful of open scans, and a small batch pays every iteration's fixed costs for little work. A pool
 * owns `slots` scan slots in HBM and runs ONE launch sequence per iteration over the union of the open scans of every job admitted so
 * far; a job = a set of scans with their initial poses, submitted at any time (its points are copied into free source regions on
 * the copy stream while the pool iterates, ahead of the slots) and collected by ticket. Every `chunk` iterations the host reads the
 * flags: finished scans leave, waiting scans enter one by one as slots come free. A job of `scans_per_job` scans gets, bit for bit, the poses locgpu_*_align_batch gives a plain batch of those
 * scans (same kernels; the partial sums are split as that batch would split them).
 * The pool matches against the target (tree or NDT voxels) the context holds when its chunks run, with the options given here.
 * Calls on a pool follow the context's rule: one caller thread. The host clouds of a submit must stay valid until
 * locgpu_pool_wait has returned for its ticket (they are packed by the context's upload service beside the caller).
 * With a communicator on the context (locgpu_comm_init) a job is SHARDED like a sharded batch — n_total scans, this rank holds the
 * points of [first_scan, first_scan + n_local) — submit and wait are collective calls made in the same order on every rank, and
 * each pooled iteration has one all-reduce of [slots][32] doubles; every rank ends with every pose. */
typedef struct locgpu_pool locgpu_pool;
typedef struct locgpu_pool_opts {
    int32_t slots;          /* scans the pool iterates at a time                                                     */
    int32_t prefetch;       /* source regions beyond `slots`: scans whose points are in HBM ahead of a free slot     */
                            /* (-1 = as many as slots). A job is accepted as soon as it has regions.                 */
    int32_t scans_per_job;  /* the job size whose plain-batch bits a job gets (0: as a batch of `slots` scans)        */
    int32_t chunk;          /* iterations between two looks at the flags, 1..8 (0 = 4)                               */
    int32_t matcher;        /* 0 = ICP with `icp`; 1 = NDT against the context's NDT target (locgpu_ndt_set_target)  */
    uint64_t max_points;    /* per scan                                                                              */
    locgpu_icp_opts icp;
} locgpu_pool_opts;
LOCGPU_API void locgpu_pool_opts_default(locgpu_pool_opts* o);
LOCGPU_API int locgpu_pool_create(locgpu_ctx* ctx, const locgpu_pool_opts* opts, locgpu_pool** out);
LOCGPU_API void locgpu_pool_destroy(locgpu_pool* pool);
/* init_poses: n_total × 7. Returns once the job has source regions (it lets running scans finish when there are none) and its copy
 * has been started; *ticket identifies it. Unsharded: n_local = n_total, first_scan = 0. */
LOCGPU_API int locgpu_pool_submit(locgpu_pool* pool, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n_local, int first_scan,
                                  int n_total, const double* init_poses, int64_t* ticket);
/* Runs the pool until every scan of the job has finished; out_poses n_total × 7, stats (optional) n_total. A ticket is good once. */
LOCGPU_API int locgpu_pool_wait(locgpu_pool* pool, int64_t ticket, double* out_poses, locgpu_align_stats* stats);
/* One turn of the pool without collecting anything: look at the chunk in flight (block != 0: wait for it; with several ranks a
 * non-blocking look is a no-op), let finished scans out and waiting jobs in, enqueue the next chunk. For callers that keep the pool
 * full — submit whenever locgpu_pool_info reports room — instead of waiting for the oldest ticket. *done: every scan of the job has
 * finished (locgpu_pool_wait returns at once). */
LOCGPU_API int locgpu_pool_step(locgpu_pool* pool, int block);
LOCGPU_API int locgpu_pool_done(const locgpu_pool* pool, int64_t ticket, int* done);
/* out = {slots, free slots, jobs not collected, pooled iterations launched, Σ over them of the open scans this rank held, open scans,
 *        source regions, free source regions} */
LOCGPU_API int locgpu_pool_info(const locgpu_pool* pool, int64_t out[8]);
/* With locgpu_profile_enable on: out[0] = device ms of the chunks run since the last reset (HIP events on the pool's stream), out[1] = chunks. */
LOCGPU_API int locgpu_pool_profile_read(locgpu_pool* pool, double out[2], int reset);

/* ---- NDT target: NdtRegistration::SetInputTarget → SetDirectNdtTargetCloud (ndt_registration.cpp:65-85, 87-148), or with
 * opts->method == 2 → SetIncNdtTargetCloud (:150-183): the voxel set then PERSISTS across calls (LRU of opts->capacity voxels,
 * statistics of a voxel recomputed from the points the latest call put into it); switching method, voxel_size or capacity,
 * or a direct call, starts from an empty set. */
LOCGPU_API int locgpu_ndt_set_target(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, const locgpu_ndt_opts* opts);
/* out[0]=voxels kept, out[1]=hash-table capacity, out[2]=bytes in HBM */
LOCGPU_API int locgpu_ndt_target_info(const locgpu_ctx* ctx, int64_t out[3]);
/* Read the voxel table back (tests): keys n×3 int32, mu n×3 f64, info n×9 f64 row-major; returns count through *n_out. */
LOCGPU_API int locgpu_ndt_dump(locgpu_ctx* ctx, int32_t* keys, double* mu, double* info, size_t cap, size_t* n_out);
/* ---- NdtRegistration::ScanMatch minus the output cloud (ndt_registration.cpp:238-257 → AlignNdt :374-464).
 * When stats->status == 1 the reference leaves result_pose unassigned; out_pose then holds init_pose. */
LOCGPU_API int locgpu_ndt_align(locgpu_ctx* ctx, const void* src, size_t n, size_t stride_bytes, const double init_pose[7],
                                double out_pose[7], locgpu_align_stats* stats);

/* ---- hipGraph mode (BASELINE config 5, streaming loop). When on, an align call replays instantiated graphs of Gauss–Newton
 * iterations instead of launching kernel by kernel: one graph of the first eight iterations (with the state upload and
 * read-back) — the typical alignment ends inside it: one graph launch, one host synchronisation — and a four-iteration graph
 * replayed while scans are still open. Kernels early-out per scan on a device-side flag, so the fixed node sequence equals the
 * reference's data-dependent loop (icp_registration.cpp:358-376). Graphs are captured once per batch, options and target and
 * re-captured when any of them changes. Results are bit-identical to the eager mode. */
LOCGPU_API int locgpu_graph_enable(locgpu_ctx* ctx, int on);

/* ---- Measurement hooks (bench.py / tests; no reference counterpart). */
/* Average device time in ms of each hot kernel over the calls since the last reset, measured with hipEvents on the
 * context's stream: out[0]=search, out[1]=fit+accumulate, out[2]=solve/update, out[3..5]=their launch counts.
 * Timing is only collected when enabled (it serialises launches with events): on = 1 times all three stages (four event
 * records per iteration), on = 2 only the search stage (two records per iteration; out[1], out[2] and their counts stay 0). */
LOCGPU_API int locgpu_profile_enable(locgpu_ctx* ctx, int on);
LOCGPU_API int locgpu_profile_read(locgpu_ctx* ctx, double out[6], int reset);
/* Total tree nodes / leaves visited by the search kernel of the NEXT align/hb call(s) when counting is on
 * (separate instrumented kernel; never on in timed runs). out[0]=nodes, out[1]=leaves, out[2]=queries, out[3]=distinct 8-byte
 * tree slots a search launch read at all, summed over the launches (the compulsory tree traffic of the search stage). */
LOCGPU_API int locgpu_visit_count_enable(locgpu_ctx* ctx, int on);
LOCGPU_API int locgpu_visit_count_read(locgpu_ctx* ctx, uint64_t out[4], int reset);

/* Search bookkeeping since the last reset (enabled by the first call): out[0] = queries handled by the fast search
 * kernel's launches, out[1] = queries it handed to the exact redo kernel (distance ties / near-misses on the top tree levels),
 * out[2] = grid mode: queries the tile kernel handed to the ring-walk kernel, out[3] = diagnostic build (LOCGPU_STAMP=1) only: queries of the
 * fast tree search that had to walk the un-stored top levels of their first descent again ("replays"); 0 otherwise. */
LOCGPU_API int locgpu_search_stats_read(locgpu_ctx* ctx, uint64_t out[4], int reset);
/* Test hook: the neighbour lists the batch's most recent search stage left behind (the one inside the last locgpu_icp_hb_batch /
 * align call on it; k = 5, or 1 for P2P), as ORIGINAL point indices of the target cloud in ascending distance — the order
 * KdTree::GetClosestPoint returns (kdtree.cpp:160-165). out[(scan * max_points + i) * k + j]; -1 where there is none
 * (points beyond a scan's count hold stale values). This is how the tests compare the HOT search kernel's lists with the oracle's. */
LOCGPU_API int locgpu_debug_batch_nn(locgpu_ctx* ctx, locgpu_batch* batch, int k, int32_t* out);

/* =====================================================================================================================
 * Clouds resident in HBM and the filters either side of the matcher.
 *
 * In the reference every scan passes RemoveNanPoint → VoxelFilter::Filter → ScanMatch (loc.cpp:217-224, lio.cpp:236,257),
 * the global map passes BoxFilter::Filter → SetInputTarget (loc.cpp:187-194) and every keyframe passes
 * pcl::transformPointCloud → operator+= → VoxelFilter::Filter → SetInputTarget (lio.cpp:268-306). All of these are thin
 * wrappers over PCL 1.8 (pcl::removeNaNFromPointCloud, pcl::VoxelGrid, pcl::CropBox, pcl::transformPointCloud). The entry
 * points below run the same steps on the GPU; a locgpu_cloud keeps a cloud in HBM between them so that a scan crosses PCIe
 * once. A cloud is {x, y, z, intensity} per point plus PCL's `is_dense` flag, which the PCL filters trust (a cloud flagged
 * dense is never tested for NaN) and which therefore travels with the data.
 *
 * Results equal PCL's: the same points in the same order (VoxelGrid: centroids in ascending voxel index; CropBox and
 * removeNaN: survivors in input order). VoxelGrid's float32 centroid sums run in input order (PCL's order inside a voxel is
 * whatever its unstable std::sort leaves, so the last bits of a centroid are not defined by PCL itself).
 * `intensity_offset` = byte offset of the float32 intensity inside a point (pcl::PointXYZI: 16), or LOCGPU_NO_INTENSITY.
 *
 * Two contexts on one GPU as a two-stage front-end (round 4): the calls of one context are strictly sequential (one caller thread,
 * like the reference's matcher), but a cloud of context A may be the SOURCE of locgpu_icp_align_cloud / locgpu_ndt_align_cloud and
 * the scan of locgpu_submap_add_keyframe on context B of the same GPU (B's stream is ordered behind what A has enqueued; A must
 * not modify the cloud while B uses it). A caller thread that uploads and filters scan i+1 on A while another matches scan i on B
 * overlaps the two stages — the poses are the sequential loop's, bit for bit (tests/test_gpu_filters.py, pipelined streaming loop).
 * ===================================================================================================================== */
typedef struct locgpu_cloud locgpu_cloud;
typedef struct locgpu_submap locgpu_submap;
#define LOCGPU_NO_INTENSITY ((size_t)-1)

LOCGPU_API int locgpu_cloud_create(locgpu_ctx* ctx, locgpu_cloud** out);
LOCGPU_API void locgpu_cloud_destroy(locgpu_cloud* c);
/* Host → HBM (deep copy, like every reference call that takes a CloudPtr). */
LOCGPU_API int locgpu_cloud_upload(locgpu_cloud* c, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int is_dense);
LOCGPU_API int locgpu_cloud_info(const locgpu_cloud* c, size_t* n, int* is_dense);
/* HBM → host: writes x, y, z (and intensity unless LOCGPU_NO_INTENSITY) of each point, leaves the other bytes of the
 * caller's points alone. Fails with LOCGPU_ERR_INVALID when capacity < the cloud's size. */
LOCGPU_API int locgpu_cloud_download(const locgpu_cloud* c, void* out, size_t capacity, size_t stride_bytes, size_t intensity_offset);
LOCGPU_API int locgpu_cloud_copy(const locgpu_cloud* in, locgpu_cloud* out);

/* RemoveNanPoint, LocUtils/include/LocUtils/common/point_cloud_utils.h:13-20 (pcl::removeNaNFromPointCloud). out may be in. */
LOCGPU_API int locgpu_cloud_remove_nan(const locgpu_cloud* in, locgpu_cloud* out);
/* VoxelFilter::Filter, LocUtils/src/model/cloud_filter/voxel_filter.cpp:19-25 (pcl::VoxelGrid, leaf = (v, v, v), defaults).
 * *passthrough (optional) = 1 when PCL's "leaf size is too small for the input dataset" rule copied the input unchanged.
 * out may be in (the reference filters local_map_ in place, lio.cpp:300). */
LOCGPU_API int locgpu_cloud_voxel_filter(const locgpu_cloud* in, float leaf, locgpu_cloud* out, int* passthrough);
/* BoxFilter::Filter, LocUtils/src/model/cloud_filter/box_filter.cpp:25-32 (pcl::CropBox, inclusive bounds). The caller
 * computes the edges as BoxFilter::CalculateEdge does (:59-66): min = origin − size, max = origin + size in float32. */
LOCGPU_API int locgpu_cloud_crop_box(const locgpu_cloud* in, const float min_xyz[3], const float max_xyz[3], locgpu_cloud* out);
/* pcl::transformPointCloud(in, out, pose.matrix()) with the double-precision matrix of lio.cpp:244,279. out may be in. */
LOCGPU_API int locgpu_cloud_transform(const locgpu_cloud* in, const double pose[7], locgpu_cloud* out);
/* pcl::PointCloud::operator+= (lio.cpp:245,291,297). */
LOCGPU_API int locgpu_cloud_append(locgpu_cloud* dst, const locgpu_cloud* src);

/* The matcher entry points on resident clouds (same semantics as their host-pointer versions above). */
LOCGPU_API int locgpu_icp_set_target_cloud(locgpu_ctx* ctx, const locgpu_cloud* target);
/* The same SetInputTarget (icp_registration.cpp:14-22; Lio re-ingests its local map every keyframe, lio.cpp:296-305) with the host
 * tree build on a worker thread: returns once the cloud has been copied out (the caller may change or free it), the build runs while
 * the caller uploads and filters the next scan, and the first entry point that reads the ICP target — an align, H/B, k-NN,
 * target_info — completes the ingest (device buffers, copy) on the caller's thread and reports its errors. Until then the previous
 * target stays in place; another SetInputTarget supersedes a pending one. Like every SetInputTarget it must not be called between a
 * locgpu_*_align_batch_begin and its end. Results are those of locgpu_icp_set_target_cloud. */
LOCGPU_API int locgpu_icp_set_target_cloud_async(locgpu_ctx* ctx, const locgpu_cloud* target);
LOCGPU_API int locgpu_ndt_set_target_cloud(locgpu_ctx* ctx, const locgpu_cloud* target, const locgpu_ndt_opts* opts);
LOCGPU_API int locgpu_icp_align_cloud(locgpu_ctx* ctx, const locgpu_cloud* src, const double init_pose[7], const locgpu_icp_opts* opts,
                                      double out_pose[7], locgpu_align_stats* stats);
LOCGPU_API int locgpu_ndt_align_cloud(locgpu_ctx* ctx, const locgpu_cloud* src, const double init_pose[7], double out_pose[7],
                                      locgpu_align_stats* stats);

/* Host-pointer one-shots (upload → filter → download), what VoxelFilter::Filter / BoxFilter::Filter / RemoveNanPoint bind to.
 * `out` needs room for n points; *out_n receives the count, *out_is_dense (optional) the flag of the result. out may be pts. */
LOCGPU_API int locgpu_voxel_filter(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int is_dense, float leaf,
                                   void* out, size_t* out_n, int* out_is_dense);
LOCGPU_API int locgpu_crop_box(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int is_dense,
                               const float min_xyz[3], const float max_xyz[3], void* out, size_t* out_n, int* out_is_dense);
LOCGPU_API int locgpu_remove_nan(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int is_dense, void* out,
                                 size_t* out_n, int* out_is_dense);

/* LoamFeatureExtract::Extract + ExtractFromSector, LocUtils/src/model/feature_extract/loam_feature_extract.cpp:19-151 (called on every
 * scan by Lio::AddCloud(FullCloudPtr), lio.cpp:323): per-ring curvature, six sectors per ring, at most 20 edge points per sector,
 * every unmarked point a surface point; outputs ring by ring, sector by sector, edges in descending and surface points in
 * ascending curvature. `ring` = one byte per point (FullPointType::ring, point_types.h:70). Equal curvatures (an order the
 * reference's std::sort leaves open) are ordered by ring position. edge and surf must be distinct clouds of in's context.
 * Limits: num_scan ≤ 256, rings of at most 6 × 2048 points. */
LOCGPU_API int locgpu_cloud_loam_extract(const locgpu_cloud* in, const uint8_t* ring, int num_scan, locgpu_cloud* edge, locgpu_cloud* surf);
/* Host-pointer one-shot on the reference's FullPointType layout (point_types.h:65-78: x,y,z at 0, uint8 intensity at 24, uint8 ring
 * at 25, stride 64): intensity_is_u8 = 1 converts like `p.intensity = pt.intensity` (:33). Both outputs need room for n points. */
LOCGPU_API int locgpu_loam_extract(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes, size_t intensity_offset, int intensity_is_u8,
                                   size_t ring_offset, int num_scan, void* edge_out, size_t* n_edge, void* surf_out, size_t* n_surf,
                                   size_t out_stride_bytes, size_t out_intensity_offset);

/* The local map of Lio::AddCloud's keyframe branch (lio.cpp:268-306), kept in HBM: a queue of at most num_kfs world-frame
 * keyframe clouds (scans_in_local_map_) and the voxel-filtered local map (local_map_) that is the next matching target.
 * add_keyframe(scan, pose): key_frame_scan = transform(scan, pose) (pose NULL: scan is already in the world frame); push;
 * when the queue exceeds num_kfs drop the oldest and rebuild the map from the retained keyframes, otherwise append the new
 * one to the (already filtered) map; then voxel-filter the map in place. The caller then hands locgpu_submap_cloud() to
 * locgpu_icp_set_target_cloud / locgpu_ndt_set_target_cloud (or the new keyframe alone for INCREMENTAL_NDT, lio.cpp:301-304). */
LOCGPU_API int locgpu_submap_create(locgpu_ctx* ctx, int num_kfs, float leaf, locgpu_submap** out);
LOCGPU_API void locgpu_submap_destroy(locgpu_submap* m);
LOCGPU_API int locgpu_submap_add_keyframe(locgpu_submap* m, const locgpu_cloud* scan, const double pose[7]);
LOCGPU_API int locgpu_submap_cloud(locgpu_submap* m, locgpu_cloud** map);          /* borrowed: owned by the submap */
LOCGPU_API int locgpu_submap_last_keyframe(locgpu_submap* m, locgpu_cloud** kf);   /* borrowed: the newest world-frame keyframe */
LOCGPU_API int locgpu_submap_info(const locgpu_submap* m, int* n_keyframes, size_t* map_points);

#ifdef __cplusplus
}
#endif
#endif /* LOCGPU_H_ */
