// loc_lib_amd/csrc/scan_pool.hip — the open-scan pool: the batched many-scans-vs-one-map mode as a continuous service.
//
// A batch alignment (locgpu_icp_align_batch) runs ITS scans from the first Gauss–Newton iteration to the last: the reference's
// loop (icp_registration.cpp:358-376, ndt_registration.cpp:393-462) stops per scan, so the late iterations of a batch hold a handful
// of open scans and leave the chip idle, and a small batch — the 32 scans one of eight ranks holds of BASELINE configs[3] — pays
// every iteration's fixed costs (five dependent launches, one traversal's latency, the solve) for an eighth of the work: 0.0204 ms
// per scan-iteration against 0.0132 for 256 scans (profiles/r04_baseline_table.json).
//
// The pool keeps `slots` scan slots in HBM (neighbour lists, partial sums, pose state — the layout of a batch of that many scans)
// plus an ARENA of source regions (slots + `prefetch` of them, one scan's points each) and runs ONE launch sequence per pooled
// iteration over the union of the open scans of every job admitted so far: search → fit/accumulate → solve over a device-side
// list of open slots; a slot says which region holds its points. Jobs (sets of scans with their initial poses) are submitted at
// any time — all a job needs is free REGIONS: its points are copied there on the copy stream while the pool iterates, ahead of
// the slots coming free. At a chunk boundary (every `chunk` iterations the host reads the flags) finished scans leave — slot and
// region are free again — and waiting scans enter ONE BY ONE, oldest job first, as slots are there: the pool stays full without
// waiting for a whole job's worth of room (round 5 measured the alternative — points copied into the slots themselves, a job
// admitted as a whole: every refill waited for the copy and the pool ran dry between generations of jobs). A scan's arithmetic is what it would be in a plain batch: the kernels are the same, a scan's blocks do the same work
// wherever the list finds them, and the split of the partial sums is the one a plain batch of `scans_per_job` scans uses — so a
// pooled job of that size returns the plain batch's poses bit for bit (tests/test_gpu_pool.py).
//
// Several GPUs (the context has a communicator): a job has n_total scans of which this rank holds [first, first + n_local); every
// rank gives the job the SAME n_total slots (the slot bookkeeping only depends on the order of the calls and on the convergence
// flags, which every rank sees for every scan), runs search and accumulate over the open slots it holds, and per pooled iteration
// ONE all-reduce of [slots][32] doubles replicates the sums (zeros from the ranks that do not hold a slot). As in sharded batches
// the owner of a scan solves it at once; the all-reduce and the replicas' solve run on the communication stream, off the critical
// path. submit and wait are then collective calls.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "batch_upload.hpp"
#include "context.hpp"
#include "launch.hpp"
#include "ndt_inc.hpp"
#include "ndt_kernels.hpp"
#include "pool_sched.hpp"

using namespace locgpu;

namespace {

struct PoolJob : PoolSchedJob {  // pool_sched.hpp: ticket, n_total, first, n_local, region[], next, remaining
    std::vector<int> counts;   // [n_local] points of the scans this rank holds
    std::vector<double> init;  // [n_total][7]
    std::vector<double> out;   // [n_total][7]
    std::vector<locgpu_align_stats> stats;
    // the copy of its points goes in pieces of kUploadPiece scans, each with its own event: the first scans of a large job enter the
    // pool while its last ones are still on their way (a 256-scan job is 472 MB: 14 ms at PCIe speed)
    std::vector<std::unique_ptr<BatchUploadState>> upl;
    // A piece of the job's copy failed: the job is FAILED, not stuck — its remaining scans enter with zero points (they run to
    // max_iteration without an update; direct NDT: det(H) = 0 at once), leave, give their slots and regions back, and locgpu_pool_wait
    // reports failed_rc for this ticket while every other job goes on. With several ranks the failure is this rank's alone and no
    // rank skips anything: the launch sequence and the collectives are those of a healthy job, so nobody waits in an all-reduce
    // for a rank that has left (the ranks that hold no failed copy see the scan end unconverged).
    int failed_rc = 0;
    std::string failed_msg;
};

constexpr int kUploadPiece = 32;

constexpr int kAccRing = 8;  // exchange buffers in rotation: one per pooled iteration of a chunk (chunk <= kAccRing)

}  // namespace

struct locgpu_pool {
    locgpu_ctx* ctx = nullptr;
    locgpu_batch* b = nullptr;  // storage: a batch of `slots` scans
    int slots = 0, chunk = 4, split_scans = 0;
    bool ndt = false;
    GnParams prm{};
    int k = 0;
    float alpha_eff = 0.f;
    locgpu_icp_opts icp{};
    bool with_comm = false;   // the context has a communicator: exchange step every iteration
    bool multi_rank = false;  // ... of more than one rank: nothing may depend on this rank's timing
    bool decoupled = false;   // the owner of a slot solves it ahead of the exchange, which runs on the communication stream
    int regions = 0;                      // source regions in the arena (>= slots)
    float4* d_arena = nullptr;            // [regions][max_n]
    std::unique_ptr<PoolSched> sched;     // pool_sched.hpp: free slots and regions, which scan sits where, who waits (pure host logic)
    int* h_src_of = nullptr;              // pinned [slots]: region of the slot's points
    int* d_src_of = nullptr;
    std::map<int64_t, PoolJob*> jobs;     // every job not yet handed back through locgpu_pool_wait
    int64_t next_ticket = 1;
    bool in_flight = false;               // a chunk (and the read-back of the states behind it) is enqueued
    int n_open = 0;                       // admitted scans not finished, as of the last look at the flags
    long long iterations = 0;             // pooled iterations launched so far
    long long scan_iterations = 0;        // Σ over them of the open scans this rank held
    int* h_list = nullptr;                // pinned [2][slots]: open slots this rank holds | open slots it does not
    int* d_list = nullptr;
    int* h_counts = nullptr;              // pinned [slots]
    unsigned char* h_owned = nullptr;     // pinned [slots]
    unsigned char* d_owned = nullptr;
    double* d_acc = nullptr;              // [kAccRing][slots][kAccW] (with a communicator)
    hipEvent_t ev_ready = nullptr, ev_reduced = nullptr;
    int n_mine = 0, n_theirs = 0;         // lengths of the two lists of the chunk in flight
    int acc_slot = 0;
    // measurement (locgpu_profile_enable on the context): device time of the chunks, HIP events on the pool's stream
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    bool timed = false;
    double chunk_ms = 0.0;
    long long chunks = 0;
    std::vector<hipEvent_t> stage_ev;     // profile mode 1: four per iteration of the chunk — search | fit+accumulate | solve (+ exchange)
    size_t stage_used = 0;
    int deferred_rc = 0;                  // a pump failure behind an accepted submit: reported by the next step / wait
    std::string deferred_msg;
};

namespace {

int pool_fail(locgpu_pool* P, int code, const std::string& msg) { return fail(P->ctx, code, msg); }

void job_free(PoolJob* j) {
    if (!j) return;
    for (auto& u : j->upl) {
        if (u->done_valid && u->done) (void)hipEventSynchronize(u->done);
        if (u->done) (void)hipEventDestroy(u->done);
    }
    delete j;
}

void init_state(PoseState& ps, const double* pose) {
    std::memset(&ps, 0, sizeof(ps));
    for (int i = 0; i < 4; ++i) ps.q[i] = pose[i];
    for (int i = 0; i < 3; ++i) ps.t[i] = pose[4 + i];
    quat_to_R(ps.q, ps.R);
}

// The chunk in flight has run (the caller synchronised the stream): finished scans hand their results to their jobs and leave.
void pool_collect(locgpu_pool* P) {
    locgpu_batch* b = P->b;
    for (int s = 0; s < P->slots; ++s) {
        PoolJob* j = static_cast<PoolJob*>(P->sched->job_of(s));
        if (!j || !b->h_state[s].done) continue;
        const PoseState& ps = b->h_state[s];
        const int i = P->sched->idx_of(s);
        if (ps.status == 1) {  // direct NDT aborted: the reference leaves result_pose unassigned; hand back init_pose (locgpu_api.hip write_results)
            for (int c = 0; c < 7; ++c) j->out[7 * i + c] = j->init[7 * i + c];
        } else {
            for (int c = 0; c < 4; ++c) j->out[7 * i + c] = ps.q[c];
            for (int c = 0; c < 3; ++c) j->out[7 * i + 4 + c] = ps.t[c];
        }
        locgpu_align_stats& st = j->stats[i];
        st.iterations = ps.iterations; st.converged = ps.converged; st.status = ps.status; st.reserved = 0;
        st.last_effective_num = ps.last_eff; st.last_dx_norm = ps.last_dx_norm;
        P->sched->finish(s);
    }
    if (P->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, P->ev_t0, P->ev_t1) == hipSuccess) { P->chunk_ms += ms; P->chunks++; }
        P->timed = false;
    }
    // per-stage times go where a batch alignment's go (locgpu_profile_read): [0] search, [1] fit + accumulate, [2] solve (and exchange)
    for (size_t i = 0; i + 3 < P->stage_used; i += 4)
        for (int st = 0; st < 3; ++st) {
            if (P->ndt && st == 0) continue;  // NDT has no search kernel
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, P->stage_ev[i + st], P->stage_ev[i + st + 1]) == hipSuccess) { P->ctx->prof_ms[st] += ms; P->ctx->prof_n[st] += 1; }
        }
    P->stage_used = 0;
    P->in_flight = false;
}

// One pooled Gauss–Newton iteration over the lists of the chunk (d_list: the open slots this rank holds, then the others).
bool pool_launch_iteration(locgpu_pool* P) {
    locgpu_ctx* ctx = P->ctx;
    locgpu_batch* b = P->b;
    hipStream_t s = b->stream;
    const int* mine = P->d_list;
    const int* theirs = P->d_list + P->slots;
    int n_partial_blocks = b->blocks_per_scan;
    auto mark = [&]() {
        if (ctx->profile != 1) return;
        while (P->stage_ev.size() <= P->stage_used) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) return;
            P->stage_ev.push_back(ev);
        }
        (void)hipEventRecord(P->stage_ev[P->stage_used++], s);
    };
    mark();
    if (P->n_mine > 0) {
        if (!P->ndt) {
            SearchArgs sa{ctx->d_tree, ctx->tree_slots * sizeof(uint64_t), ctx->depth, P->d_arena, b->d_counts, b->d_state, b->d_nn, b->pitch, b->max_n, b->n_scans, P->k,
                          P->alpha_eff, P->prm.method == LOCGPU_P2P ? 1 : 0, nullptr, b->d_redo_list, b->d_redo_count, b->d_redo_list2, b->d_redo_count + 1,
                          ctx->d_search_stats};
            sa.active = mine; sa.n_active = P->n_mine; sa.src_of = P->d_src_of;
            if (!ctx->tree_bounded) sa.redo_list = nullptr;  // huge / non-finite map coordinates: exact tree kernel only
            if (!launch_icp_search(sa, s)) { fail(ctx, LOCGPU_ERR_DEPTH, "pool: unsupported k/depth"); return false; }
            mark();
            const double gate = P->prm.method == LOCGPU_P2PLANE ? P->prm.max_plane_distance : (P->prm.method == LOCGPU_P2LINE ? P->prm.max_line_distance : P->prm.max_nn_distance);
            AccumArgs aa{ctx->d_tree, P->d_arena, b->d_counts, b->d_state, b->d_nn, b->pitch, b->max_n, b->n_scans, gate, b->d_partials};
            aa.active = mine; aa.n_active = P->n_mine; aa.split_scans = P->split_scans; aa.src_of = P->d_src_of;
            n_partial_blocks = launch_icp_accum(P->prm.method, aa, s);
        } else if (P->prm.method == 4) {
            mark();
            launch_inc_accum(ctx->inc, ctx->ndt_opts.res_outlier_th, ctx->ndt_opts.nearby_type == 0 ? 1 : 7, P->d_arena, b->d_counts, b->d_state, b->max_n, b->n_scans, b->d_partials, s,
                             mine, P->n_mine, P->d_src_of);
        } else {
            mark();
            n_partial_blocks = launch_ndt_accum(ctx->ndt, P->d_arena, b->d_counts, b->d_state, b->max_n, b->n_scans, b->d_partials, s, mine, P->n_mine, P->split_scans, P->d_src_of);
        }
    } else {
        mark();  // nothing local: this rank only takes part in the exchange below
    }
    mark();
    unsigned int* list_counts = P->ndt ? nullptr : b->d_redo_count;
    if (!P->with_comm) {
        launch_gn_solve(b->d_partials, n_partial_blocks, b->d_state, P->n_mine, P->prm, 1, nullptr, list_counts, s, mine);
        mark();
        return hip_ok(ctx, hipGetLastError(), "pool: kernel launch");
    }
    // The exchange step: per slot 28 sums, zeros from the ranks that do not hold it, in exactly the order
    // gn_solve_kernel would sum the block partials — all-reduce, then a solve on the reduced sums gives the one-GPU bits.
    double* acc = P->d_acc + (size_t)(P->acc_slot % kAccRing) * P->slots * kAccW;
    P->acc_slot++;
    launch_sum_partials(b->d_partials, n_partial_blocks, b->d_state, 0, P->slots, P->slots, acc, s, P->d_owned);
    if (P->decoupled) {
        // A slot's sums are complete on the rank that holds it: the owner solves at once and goes on to the next search, the
        // all-reduce and the replicas' solve follow on the communication stream (locgpu_api.hip, sharded batches).
        hipStream_t cs = ctx->comm_stream;
        if (P->n_mine > 0) launch_gn_solve(acc, 1, b->d_state, P->n_mine, P->prm, 1, nullptr, list_counts, s, mine);
        if (!hip_ok(ctx, hipEventRecord(P->ev_ready, s), "pool: hipEventRecord") || !hip_ok(ctx, hipStreamWaitEvent(cs, P->ev_ready, 0), "pool: hipStreamWaitEvent")) return false;
        if (!comm_all_reduce_f64(ctx, acc, (size_t)P->slots * kAccW, cs)) return false;
        if (P->n_theirs > 0) launch_gn_solve(acc, 1, b->d_state, P->n_theirs, P->prm, 1, nullptr, nullptr, cs, theirs);
    } else {
        // everything on the pool's stream: sums → all-reduce → every rank solves every open slot
        if (!comm_all_reduce_f64(ctx, acc, (size_t)P->slots * kAccW, s)) return false;
        if (P->n_mine > 0) launch_gn_solve(acc, 1, b->d_state, P->n_mine, P->prm, 1, nullptr, list_counts, s, mine);
        if (P->n_theirs > 0) launch_gn_solve(acc, 1, b->d_state, P->n_theirs, P->prm, 1, nullptr, nullptr, s, theirs);
    }
    mark();
    return hip_ok(ctx, hipGetLastError(), "pool: kernel launch");
}

// Admit what waits, then enqueue the next chunk over the open slots. Nothing in flight when called.
int pool_launch(locgpu_pool* P) {
    locgpu_ctx* ctx = P->ctx;
    locgpu_batch* b = P->b;
    hipStream_t s = b->stream;
    int open_now = 0;  // scans still running (the flags are fresh: the caller has just read them, or nothing has run since)
    for (int sl = 0; sl < P->slots; ++sl) open_now += (P->sched->job_of(sl) && !b->h_state[sl].done) ? 1 : 0;
    // waiting scans enter one by one, oldest job first, while there are free slots (PoolSched::admit)
    std::vector<PoolAdmitted> admitted;
    int rc_admit = LOCGPU_OK;
    P->sched->admit(
        [&](PoolSchedJob* sj, int i) {
            PoolJob* j = static_cast<PoolJob*>(sj);
            if (!j->holds(i) || (i - j->first) % kUploadPiece != 0) return true;
            // the first scan of an upload piece: a copy still on its way must not stall the scans that are running (one rank
            // only: with several ranks every decision has to be the same everywhere, so the stream simply waits for the copy)
            if (j->failed_rc) return true;  // a failed job's scans enter empty (see PoolJob::failed_rc)
            BatchUploadState* u = j->upl[(size_t)(i - j->first) / kUploadPiece].get();
            const bool may_defer = !P->multi_rank && (open_now > 0 || !admitted.empty());
            if (may_defer && upload_host_busy(u)) return false;
            const int urc = upload_join_state(ctx, u);
            if (urc != LOCGPU_OK) {
                j->failed_rc = urc;
                j->failed_msg = ctx->err;
                for (int& c : j->counts) c = 0;  // scans already in the pool keep their points (their pieces landed); the rest run on nothing
                return true;
            }
            if (may_defer && hipEventQuery(u->done) == hipErrorNotReady) return false;
            if (!hip_ok(ctx, hipStreamWaitEvent(s, u->done, 0), "pool: hipStreamWaitEvent")) { rc_admit = LOCGPU_ERR_NO_DEVICE; return false; }
            return true;
        },
        admitted);
    if (rc_admit != LOCGPU_OK) return rc_admit;  // a stream error (not a job's copy: that fails its job, above)
    for (const PoolAdmitted& a : admitted) {
        PoolJob* j = static_cast<PoolJob*>(a.job);
        init_state(b->h_state[a.slot], &j->init[7 * (size_t)a.idx]);
        const bool mine = j->holds(a.idx);
        P->h_owned[a.slot] = mine ? 1 : 0;
        P->h_counts[a.slot] = mine ? j->counts[(size_t)(a.idx - j->first)] : 0;
        P->h_src_of[a.slot] = j->region[(size_t)a.idx];
    }
    const bool admitted_any = !admitted.empty();
    int n_mine = 0, n_theirs = 0;
    for (int sl = 0; sl < P->slots; ++sl) {
        if (!P->sched->job_of(sl) || b->h_state[sl].done) continue;
        if (P->h_owned[sl]) P->h_list[n_mine++] = sl;
        else P->h_list[P->slots + n_theirs++] = sl;
    }
    P->n_mine = n_mine; P->n_theirs = n_theirs;
    P->n_open = n_mine + n_theirs;
    static const bool dbg = getenv("LOCGPU_POOL_DEBUG") != nullptr;  // diagnostic: one line per chunk
    if (dbg) {
        static const auto t0 = std::chrono::steady_clock::now();
        fprintf(stderr, "[pool] t=%.3f ms open=%d (mine %d) waiting=%zu free slots=%zu regions=%zu jobs=%zu\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                P->n_open, n_mine, P->sched->waiting(), (size_t)P->sched->free_slots(), (size_t)P->sched->free_regions(), P->jobs.size());
    }
    if (P->n_open == 0) return LOCGPU_OK;
    if (admitted_any) {
        // the host's copy of the states is the device's (read back behind the last chunk) plus the new scans' initial poses
        LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_state, b->h_state, (size_t)P->slots * sizeof(PoseState), hipMemcpyHostToDevice, s));
        LOCGPU_HIP(ctx, hipMemcpyAsync(b->d_counts, P->h_counts, (size_t)P->slots * sizeof(int), hipMemcpyHostToDevice, s));
        LOCGPU_HIP(ctx, hipMemcpyAsync(P->d_src_of, P->h_src_of, (size_t)P->slots * sizeof(int), hipMemcpyHostToDevice, s));
        if (P->d_owned) LOCGPU_HIP(ctx, hipMemcpyAsync(P->d_owned, P->h_owned, (size_t)P->slots, hipMemcpyHostToDevice, s));
    }
    if (n_mine) LOCGPU_HIP(ctx, hipMemcpyAsync(P->d_list, P->h_list, (size_t)n_mine * sizeof(int), hipMemcpyHostToDevice, s));
    if (n_theirs) LOCGPU_HIP(ctx, hipMemcpyAsync(P->d_list + P->slots, P->h_list + P->slots, (size_t)n_theirs * sizeof(int), hipMemcpyHostToDevice, s));
    P->timed = ctx->profile != 0 && P->ev_t0 && P->ev_t1 && hipEventRecord(P->ev_t0, s) == hipSuccess;
    for (int c = 0; c < P->chunk; ++c) {
        if (!pool_launch_iteration(P)) return LOCGPU_ERR_NO_DEVICE;
        P->iterations++;
        P->scan_iterations += n_mine;  // an upper bound inside a chunk (scans that finish early cost nothing but are counted)
    }
    if (P->decoupled) {  // the states of the slots other ranks hold are written on the communication stream
        LOCGPU_HIP(ctx, hipEventRecord(P->ev_reduced, ctx->comm_stream));
        LOCGPU_HIP(ctx, hipStreamWaitEvent(s, P->ev_reduced, 0));
    }
    if (P->timed) P->timed = hipEventRecord(P->ev_t1, s) == hipSuccess;
    LOCGPU_HIP(ctx, hipMemcpyAsync(b->h_state, b->d_state, (size_t)P->slots * sizeof(PoseState), hipMemcpyDeviceToHost, s));
    P->in_flight = true;
    return LOCGPU_OK;
}

// One turn of the pool: look at the chunk in flight (wait for it when `block`), let finished scans out and waiting jobs in, enqueue
// the next chunk. *progress = the flags were looked at or a chunk was enqueued.
int pool_pump(locgpu_pool* P, bool block, bool* progress = nullptr) {
    locgpu_ctx* ctx = P->ctx;
    if (progress) *progress = false;
    if (P->in_flight) {
        if (!block) {
            if (P->multi_rank) return LOCGPU_OK;  // no polling where ranks must agree
            if (hipStreamQuery(P->b->stream) == hipErrorNotReady) return LOCGPU_OK;
        }
        LOCGPU_HIP(ctx, hipStreamSynchronize(P->b->stream));
        pool_collect(P);
        if (progress) *progress = true;
    }
    const int rc = pool_launch(P);
    if (rc != LOCGPU_OK) { (void)hipStreamSynchronize(P->b->stream); return rc; }
    if (progress && P->in_flight) *progress = true;
    return LOCGPU_OK;
}

}  // namespace

extern "C" {

void locgpu_pool_opts_default(locgpu_pool_opts* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->slots = 256;
    o->prefetch = -1;
    o->scans_per_job = 32;
    o->chunk = 4;
    o->matcher = 0;
    o->max_points = 64 * 1800;
    locgpu_icp_opts_default(&o->icp);
}

int locgpu_pool_create(locgpu_ctx* ctx, const locgpu_pool_opts* o, locgpu_pool** out) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!o || !out) return fail(ctx, LOCGPU_ERR_INVALID, "pool_create: bad arguments");
    *out = nullptr;
    if (o->slots < 1 || o->slots > 65535 || o->prefetch < -1 || o->prefetch > 1000000 || o->max_points == 0 || o->chunk < 0 || o->chunk > kAccRing || o->scans_per_job < 0 || (o->matcher != 0 && o->matcher != 1))
        return fail(ctx, LOCGPU_ERR_INVALID, "pool_create: bad options (1 <= slots <= 65535, chunk <= 8, matcher 0 | 1)");
    auto* P = new locgpu_pool();
    P->ctx = ctx;
    P->slots = o->slots;
    P->chunk = o->chunk > 0 ? o->chunk : 4;
    P->split_scans = o->scans_per_job > 0 ? o->scans_per_job : o->slots;
    P->ndt = o->matcher == 1;
    P->icp = o->icp;
    int rc = P->ndt ? check_ndt(ctx, P->prm) : check_icp(ctx, &o->icp, P->prm, P->k, P->alpha_eff);
    if (rc == LOCGPU_OK && !P->ndt && P->alpha_eff < 0.f) rc = fail(ctx, LOCGPU_ERR_INVALID, "pool_create: the grid search is not available in a pool");
    if (rc == LOCGPU_OK && P->ndt) P->alpha_eff = 1.0f;
    if (rc != LOCGPU_OK) { delete P; return rc; }
    P->with_comm = ctx->comm != nullptr;
    P->multi_rank = P->with_comm && ctx->comm_world > 1;
    // LOCGPU_SHARD_DECOUPLED=0|1 (tests; the switch of the sharded batches): force the exchange onto the pool's stream / behind the owner's solve
    static const int decouple_env = [] { const char* e = getenv("LOCGPU_SHARD_DECOUPLED"); return e ? atoi(e) : -1; }();
    P->decoupled = P->with_comm && (decouple_env >= 0 ? decouple_env != 0 : P->multi_rank);
    rc = alloc_batch(ctx, o->slots, (size_t)o->max_points, &P->b);
    if (rc != LOCGPU_OK) { delete P; return rc; }
    const size_t S = (size_t)P->slots;
    // the points live in the arena, not in the storage batch's own source array
    if (P->b->d_src) { (void)hipFree(P->b->d_src); P->b->d_src = nullptr; }
    P->regions = P->slots + (o->prefetch >= 0 ? o->prefetch : P->slots);
    if (!hip_ok(ctx, hipMalloc((void**)&P->d_arena, (size_t)P->regions * P->b->max_n * sizeof(float4)), "pool: hipMalloc source arena")) { locgpu_pool_destroy(P); return LOCGPU_ERR_OOM; }
    bool ok = hip_ok(ctx, hipHostMalloc((void**)&P->h_list, 2 * S * sizeof(int)), "pool: hipHostMalloc") &&
              hip_ok(ctx, hipMalloc((void**)&P->d_list, 2 * S * sizeof(int)), "pool: hipMalloc") &&
              hip_ok(ctx, hipHostMalloc((void**)&P->h_counts, S * sizeof(int)), "pool: hipHostMalloc") &&
              hip_ok(ctx, hipHostMalloc((void**)&P->h_owned, S), "pool: hipHostMalloc") &&
              hip_ok(ctx, hipHostMalloc((void**)&P->h_src_of, S * sizeof(int)), "pool: hipHostMalloc") &&
              hip_ok(ctx, hipMalloc((void**)&P->d_src_of, S * sizeof(int)), "pool: hipMalloc") &&
              hip_ok(ctx, hipEventCreate(&P->ev_t0), "pool: hipEventCreate") && hip_ok(ctx, hipEventCreate(&P->ev_t1), "pool: hipEventCreate");
    if (ok && P->with_comm)
        ok = hip_ok(ctx, hipMalloc((void**)&P->d_owned, S), "pool: hipMalloc") &&
             hip_ok(ctx, hipMalloc((void**)&P->d_acc, (size_t)kAccRing * S * kAccW * sizeof(double)), "pool: hipMalloc") &&
             hip_ok(ctx, hipEventCreateWithFlags(&P->ev_ready, hipEventDisableTiming), "pool: hipEventCreate") &&
             hip_ok(ctx, hipEventCreateWithFlags(&P->ev_reduced, hipEventDisableTiming), "pool: hipEventCreate");
    if (!ok) { locgpu_pool_destroy(P); return LOCGPU_ERR_OOM; }
    std::memset(P->h_counts, 0, S * sizeof(int));
    std::memset(P->h_owned, 0, S);
    std::memset(P->h_src_of, 0, S * sizeof(int));
    if (!hip_ok(ctx, hipMemsetAsync(P->d_src_of, 0, S * sizeof(int), P->b->stream), "pool: hipMemset") || !hip_ok(ctx, hipStreamSynchronize(P->b->stream), "pool: hipMemset")) { locgpu_pool_destroy(P); return LOCGPU_ERR_NO_DEVICE; }
    for (size_t s = 0; s < S; ++s) { std::memset(&P->b->h_state[s], 0, sizeof(PoseState)); P->b->h_state[s].done = 1; }  // a free slot is a finished scan
    P->sched.reset(new PoolSched(P->slots, P->regions));
    *out = P;
    return LOCGPU_OK;
}

void locgpu_pool_destroy(locgpu_pool* P) {
    if (!P) return;
    locgpu_ctx* ctx = P->ctx;
    (void)hipSetDevice(ctx->device);
    upload_drain(ctx);  // no job's scans are being packed into the pool's slots any more
    if (P->b) (void)hipStreamSynchronize(P->b->stream);
    if (P->with_comm && ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    for (auto& kv : P->jobs) job_free(kv.second);
    if (P->h_list) (void)hipHostFree(P->h_list);
    if (P->d_list) (void)hipFree(P->d_list);
    if (P->h_counts) (void)hipHostFree(P->h_counts);
    if (P->h_owned) (void)hipHostFree(P->h_owned);
    if (P->h_src_of) (void)hipHostFree(P->h_src_of);
    if (P->d_src_of) (void)hipFree(P->d_src_of);
    if (P->d_arena) (void)hipFree(P->d_arena);
    if (P->d_owned) (void)hipFree(P->d_owned);
    if (P->d_acc) (void)hipFree(P->d_acc);
    for (hipEvent_t ev : P->stage_ev) (void)hipEventDestroy(ev);
    if (P->ev_t0) (void)hipEventDestroy(P->ev_t0);
    if (P->ev_t1) (void)hipEventDestroy(P->ev_t1);
    if (P->ev_ready) (void)hipEventDestroy(P->ev_ready);
    if (P->ev_reduced) (void)hipEventDestroy(P->ev_reduced);
    if (P->b) free_batch(P->b);
    delete P;
}

int locgpu_pool_submit(locgpu_pool* P, const void* const* srcs, const size_t* counts, size_t stride_bytes, int n_local, int first_scan, int n_total,
                       const double* init_poses, int64_t* ticket) {
    if (!P) return LOCGPU_ERR_INVALID;
    locgpu_ctx* ctx = P->ctx;
    if (!ticket || !init_poses || n_total < 1 || n_local < 0 || first_scan < 0 || first_scan + n_local > n_total || (n_local > 0 && (!srcs || !counts)) || stride_bytes < 12)
        return pool_fail(P, LOCGPU_ERR_INVALID, "pool_submit: bad arguments");
    if (n_total > P->regions) return pool_fail(P, LOCGPU_ERR_INVALID, "pool_submit: the job has more scans than the pool has source regions (slots + prefetch)");
    if (!P->with_comm && n_local != n_total)
        return pool_fail(P, LOCGPU_ERR_INVALID, "pool_submit: this rank holds only part of the job and locgpu_comm_init has not been called");
    for (int i = 0; i < n_local; ++i) {
        if (counts[i] > (size_t)P->b->max_n) return pool_fail(P, LOCGPU_ERR_INVALID, "pool_submit: a scan has more points than the pool was created for");
        if (counts[i] && !srcs[i]) return pool_fail(P, LOCGPU_ERR_INVALID, "pool_submit: NULL scan pointer");
    }
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    // the target the pool was created against is still the context's (a pool does not outlive a SetInputTarget)
    {
        GnParams prm{};
        int k = 0;
        float a = 0.f;
        const int rc = P->ndt ? check_ndt(ctx, prm) : check_icp(ctx, &P->icp, prm, k, a);
        if (rc != LOCGPU_OK) return rc;
    }
    // room in the arena: let scans finish. Every rank accounts n_total regions for the job, whichever scans it holds: the calls a
    // rank makes must not depend on the shard it happens to hold.
    while (P->sched->free_regions() < n_total) {
        bool progress = false;
        const int rc = pool_pump(P, true, &progress);
        if (rc != LOCGPU_OK) return rc;
        if (!progress) return pool_fail(P, LOCGPU_ERR_INVALID, "pool_submit: no free source regions and nothing running");
    }
    auto* j = new PoolJob();
    j->ticket = P->next_ticket++;
    j->n_total = n_total; j->first = first_scan; j->n_local = n_local;
    if (!P->sched->accept(j)) { job_free(j); return pool_fail(P, LOCGPU_ERR_INVALID, "pool_submit: no free source regions"); }
    j->counts.resize(n_local);
    for (int i = 0; i < n_local; ++i) j->counts[i] = (int)counts[i];
    j->init.assign(init_poses, init_poses + 7 * (size_t)n_total);
    j->out.assign(7 * (size_t)n_total, 0.0);
    j->stats.assign(n_total, locgpu_align_stats{});
    int rc_up = LOCGPU_OK;
    for (int lo = 0; lo < n_local && rc_up == LOCGPU_OK; lo += kUploadPiece) {
        const int cnt = std::min(kUploadPiece, n_local - lo);
        j->upl.emplace_back(new BatchUploadState());
        rc_up = upload_start_regions(P->b, j->upl.back().get(), P->d_arena, P->regions, srcs + lo, counts + lo, stride_bytes, cnt, j->region.data() + first_scan + lo);
    }
    if (n_local > 0) {
        const int rc = rc_up;
        if (rc != LOCGPU_OK) {
            upload_drain(ctx);  // the pieces already queued read the caller's clouds and write regions that go back to the pool
            P->sched->cancel_last(j);
            job_free(j);
            return rc;
        }
    }
    P->jobs[j->ticket] = j;
    *ticket = j->ticket;
    // keep the pool turning while the caller only submits: an idle pool starts at once; a running one is looked at without waiting.
    // The job is LIVE from here on (its copy reads the caller's clouds, its ticket must be waited for): a failure of this turn of the
    // pool is not the submit's — it is kept and returned by the next locgpu_pool_step / locgpu_pool_wait.
    const int prc = pool_pump(P, false);
    if (prc != LOCGPU_OK && P->deferred_rc == LOCGPU_OK) { P->deferred_rc = prc; P->deferred_msg = ctx->err; }
    return LOCGPU_OK;
}

int locgpu_pool_wait(locgpu_pool* P, int64_t ticket, double* out_poses, locgpu_align_stats* stats) {
    if (!P) return LOCGPU_ERR_INVALID;
    locgpu_ctx* ctx = P->ctx;
    auto it = P->jobs.find(ticket);
    if (it == P->jobs.end() || !out_poses) return pool_fail(P, LOCGPU_ERR_INVALID, "pool_wait: unknown ticket or NULL output");
    PoolJob* j = it->second;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    if (P->deferred_rc != LOCGPU_OK) { const int rc = P->deferred_rc; P->deferred_rc = LOCGPU_OK; return pool_fail(P, rc, P->deferred_msg); }
    while (j->remaining > 0) {
        bool progress = false;
        const int rc = pool_pump(P, true, &progress);
        if (rc != LOCGPU_OK) return rc;
        if (!progress && j->remaining > 0) return pool_fail(P, LOCGPU_ERR_INVALID, "pool_wait: the pool has stopped with the job unfinished");
    }
    std::memcpy(out_poses, j->out.data(), j->out.size() * sizeof(double));
    if (stats) std::memcpy(stats, j->stats.data(), j->stats.size() * sizeof(locgpu_align_stats));
    int rc = LOCGPU_OK;
    if (j->failed_rc) {
        // the pieces that did not fail may still be read by the upload service: the caller frees its clouds when this returns
        for (auto& u : j->upl) (void)upload_join_state(ctx, u.get());
        rc = pool_fail(P, j->failed_rc, "pool: the copy of job " + std::to_string((long long)ticket) + " failed (" + j->failed_msg + "); its poses are not valid");
    }
    P->jobs.erase(it);
    job_free(j);
    return rc;
}

int locgpu_pool_step(locgpu_pool* P, int block) {
    if (!P) return LOCGPU_ERR_INVALID;
    LOCGPU_HIP(P->ctx, hipSetDevice(P->ctx->device));
    if (P->deferred_rc != LOCGPU_OK) { const int rc = P->deferred_rc; P->deferred_rc = LOCGPU_OK; return pool_fail(P, rc, P->deferred_msg); }
    return pool_pump(P, block != 0);
}

int locgpu_pool_done(const locgpu_pool* P, int64_t ticket, int* done) {
    if (!P || !done) return LOCGPU_ERR_INVALID;
    auto it = P->jobs.find(ticket);
    if (it == P->jobs.end()) return fail(P->ctx, LOCGPU_ERR_INVALID, "pool_done: unknown ticket");
    *done = it->second->remaining == 0;
    return LOCGPU_OK;
}

int locgpu_pool_profile_read(locgpu_pool* P, double out[2], int reset) {
    if (!P || !out) return LOCGPU_ERR_INVALID;
    out[0] = P->chunk_ms;
    out[1] = (double)P->chunks;
    if (reset) { P->chunk_ms = 0.0; P->chunks = 0; }
    return LOCGPU_OK;
}

int locgpu_pool_info(const locgpu_pool* P, int64_t out[8]) {
    if (!P || !out) return LOCGPU_ERR_INVALID;
    out[6] = P->regions;
    out[7] = P->sched ? P->sched->free_regions() : 0;
    out[0] = P->slots;
    out[1] = P->sched ? P->sched->free_slots() : 0;
    out[2] = (int64_t)P->jobs.size();
    out[3] = P->iterations;
    out[4] = P->scan_iterations;
    out[5] = P->n_open;
    return LOCGPU_OK;
}

}  // extern "C"
