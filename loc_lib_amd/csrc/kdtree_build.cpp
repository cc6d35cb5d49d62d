// loc_lib_amd/csrc/kdtree_build.cpp
//
// Host-side ingest of the ICP target: builds the reference's mean-split KD-tree
// (LocUtils/src/model/search_point/kdtree/kdtree.cpp:10-31 BuildTree, :58-94 Insert,
// :96-123 FindSplitAxisAndThresh; mean/variance per LocUtils/include/LocUtils/common/math_utils.h:35-47)
// directly into the packed, pointer-free layout the traversal kernel streams from HBM.
//
// Why on the host: split thresholds are float32 means accumulated SEQUENTIALLY in index order; any
// reassociation changes a threshold by an ulp and with it the tree. Sub-trees are independent, so
// the build is parallel across sub-trees while every node keeps the reference's summation order.
// This runs once per SetInputTarget (map change), never per scan.
//
// Packed layout (8-byte slots, preorder):
//   internal node  : 1 slot  { float thresh ; u32 (axis<<30) | right_child_slot }   left child = slot+1
//   leaf           : 2 slots { float x ; u32 (3<<30) | point_index } { float y ; float z }
// A 16-byte load at a leaf's slot therefore returns the whole point.
#include "kdtree_build.hpp"

#include <unistd.h>
#if defined(__SSE2__) && !defined(LOCGPU_SCALAR_VEC)
#include <immintrin.h>
#else
// No SSE (an aarch64 host; or -DLOCGPU_SCALAR_VEC: the CPU suite builds this branch too): the handful of intrinsics below, lane by
// lane. The reference's sums are scalar float32 adds in index order (math_utils.h:40-45); an SSE lane does exactly that, and so does
// this — the tree is the same tree.
#include <cstdint>
#include <cstring>
struct alignas(16) __m128 { float v[4]; };
struct alignas(16) __m128i { int32_t v[4]; };
static inline __m128 _mm_setzero_ps() { return __m128{{0.f, 0.f, 0.f, 0.f}}; }
static inline __m128 _mm_set1_ps(float a) { return __m128{{a, a, a, a}}; }
static inline __m128 _mm_load_ps(const float* p) { __m128 r; std::memcpy(r.v, p, 16); return r; }
static inline void _mm_store_ps(float* p, __m128 a) { std::memcpy(p, a.v, 16); }
#define LOCGPU_LANEWISE(name, op) static inline __m128 name(__m128 a, __m128 b) { __m128 r; for (int i = 0; i < 4; ++i) r.v[i] = a.v[i] op b.v[i]; return r; }
LOCGPU_LANEWISE(_mm_add_ps, +) LOCGPU_LANEWISE(_mm_sub_ps, -) LOCGPU_LANEWISE(_mm_mul_ps, *) LOCGPU_LANEWISE(_mm_div_ps, /)
#undef LOCGPU_LANEWISE
static inline __m128i _mm_set_epi32(int e3, int e2, int e1, int e0) { return __m128i{{e0, e1, e2, e3}}; }
static inline __m128i _mm_set1_epi32(int a) { return __m128i{{a, a, a, a}}; }
static inline __m128 _mm_castsi128_ps(__m128i a) { __m128 r; std::memcpy(r.v, a.v, 16); return r; }
static inline __m128 _mm_and_ps(__m128 a, __m128 b) { uint32_t x[4], y[4]; std::memcpy(x, a.v, 16); std::memcpy(y, b.v, 16); for (int i = 0; i < 4; ++i) x[i] &= y[i]; __m128 r; std::memcpy(r.v, x, 16); return r; }
static inline __m128 _mm_andnot_ps(__m128 a, __m128 b) { uint32_t x[4], y[4]; std::memcpy(x, a.v, 16); std::memcpy(y, b.v, 16); for (int i = 0; i < 4; ++i) x[i] = ~x[i] & y[i]; __m128 r; std::memcpy(r.v, x, 16); return r; }
static inline void _mm_pause() {}
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

namespace locgpu {
namespace {

struct Local {  // per-task counters (shared atomics on every leaf cost more than the build itself on small maps)
    int64_t leaves = 0;
    int depth = 0;
};

// A point travels with its index: every node's points are CONTIGUOUS 16-byte records in the reference's order (the stable
// partition keeps it), so each pass streams its range instead of gathering through an index list.
struct alignas(16) Rec { float x, y, z; int32_t id; };

struct Builder {
    std::atomic<int64_t> leaves{0};
    std::atomic<int> depth{0};
    void merge(const Local& l) {
        leaves.fetch_add(l.leaves, std::memory_order_relaxed);
        int d = depth.load(std::memory_order_relaxed);
        while (l.depth > d && !depth.compare_exchange_weak(d, l.depth, std::memory_order_relaxed)) {}
    }

    // Split a[0, len) the way FindSplitAxisAndThresh does (kdtree.cpp:96-123; mean and variance as math_utils.h:35-47: float32 sums
    // in index order, one IEEE operation at a time — the three coordinates ride in one SSE register, lane by lane the scalar code).
    // `sum_in` (optional): Σ of the points, already accumulated by the parent's partition — a child's points are appended in their
    // final order there, so its sum costs no pass of its own. Leaves the children's sums in sum_l / sum_r the same way (the side a
    // point does not go to adds +0.0f, which changes no float32 sum: a sum that starts at +0 is never −0).
    // Returns false for the degenerate case (every point on one side).
    bool split(Rec* a, Rec* tmp, size_t len, const __m128* sum_in, int& axis, float& th, size_t& n_left, __m128& sum_l, __m128& sum_r) const {
        const __m128 xyz_mask = _mm_castsi128_ps(_mm_set_epi32(0, -1, -1, -1));
        __m128 s;
        if (sum_in) {
            s = *sum_in;
        } else {
            s = _mm_setzero_ps();
            for (size_t i = 0; i < len; ++i) s = _mm_add_ps(s, _mm_and_ps(_mm_load_ps(&a[i].x), xyz_mask));
        }
        const __m128 m = _mm_div_ps(s, _mm_set1_ps((float)len));
        __m128 v = _mm_setzero_ps();
        for (size_t i = 0; i < len; ++i) {
            const __m128 d = _mm_sub_ps(_mm_and_ps(_mm_load_ps(&a[i].x), xyz_mask), m);
            v = _mm_add_ps(v, _mm_mul_ps(d, d));
        }
        v = _mm_div_ps(v, _mm_set1_ps((float)(len - 1)));
        alignas(16) float mv[4], vv[4];
        _mm_store_ps(mv, m);
        _mm_store_ps(vv, v);
        axis = 0;
        float best = vv[0];
        if (vv[1] > best) { best = vv[1]; axis = 1; }
        if (vv[2] > best) { best = vv[2]; axis = 2; }
        th = mv[axis];
        size_t nl = 0, nr = 0;
        __m128 sl = _mm_setzero_ps(), sr = _mm_setzero_ps();
        for (size_t i = 0; i < len; ++i) {  // stable partition: `< th` left, else right; both targets are written, one counter moves
            const Rec r = a[i];
            const __m128 p = _mm_and_ps(_mm_load_ps(&a[i].x), xyz_mask);
            const float c = axis == 0 ? r.x : (axis == 1 ? r.y : r.z);
            const bool left = c < th;
            const __m128 lm = _mm_castsi128_ps(_mm_set1_epi32(left ? -1 : 0));
            sl = _mm_add_ps(sl, _mm_and_ps(p, lm));
            sr = _mm_add_ps(sr, _mm_andnot_ps(lm, p));
            a[nl] = r;      // nl <= i: the record at i has been read
            tmp[nr] = r;
            nl += left;
            nr += !left;
        }
        if (nr <= 8) { for (size_t i = 0; i < nr; ++i) a[nl + i] = tmp[i]; }
        else std::memcpy(a + nl, tmp, nr * sizeof(Rec));
        n_left = nl;
        sum_l = sl;
        sum_r = sr;
        return !(nl == 0 || nr == 0);
    }

    static void emit_leaf(uint64_t* out, size_t& n_out, const Rec& r, Local& loc) {
        uint32_t xb, yb, zb;
        std::memcpy(&xb, &r.x, 4); std::memcpy(&yb, &r.y, 4); std::memcpy(&zb, &r.z, 4);
        out[n_out++] = (uint64_t)xb | ((uint64_t)((3u << 30) | (uint32_t)r.id) << 32);
        out[n_out++] = (uint64_t)yb | ((uint64_t)zb << 32);
        loc.leaves++;
    }

    void note_depth(int level) {
        int d = depth.load(std::memory_order_relaxed);
        while (level > d && !depth.compare_exchange_weak(d, level, std::memory_order_relaxed)) {}
    }

    // Recursive build of one sub-tree: slots go to out[n_out…] (room for 3·len − 1), child indices are positions in `out`; the slot
    // position of every leaf goes to leaf_out[n_leaf…] when leaf_out is given (the caller then passes the WHOLE tree's arrays and
    // the sub-tree's first slot / leaf in n_out / n_leaf).
    void build(Rec* a, Rec* tmp, size_t len, const __m128* sum_in, int level, uint64_t* out, size_t& n_out, uint32_t* leaf_out, size_t& n_leaf, Local& loc) {
        if (level > loc.depth) loc.depth = level;
        if (len == 1) { if (leaf_out) leaf_out[n_leaf++] = (uint32_t)n_out; emit_leaf(out, n_out, a[0], loc); return; }
        const Rec first = a[0];  // points[0]: the leaf of the degenerate case (kdtree.cpp:66-70)
        int axis; float th; size_t nl;
        __m128 sl, sr;
        if (!split(a, tmp, len, sum_in, axis, th, nl, sl, sr)) { if (leaf_out) leaf_out[n_leaf++] = (uint32_t)n_out; emit_leaf(out, n_out, first, loc); return; }
        const size_t pos = n_out++;
        build(a, tmp, nl, &sl, level + 1, out, n_out, leaf_out, n_leaf, loc);
        const size_t right = n_out;
        uint32_t tb; std::memcpy(&tb, &th, 4);
        out[pos] = (uint64_t)tb | ((uint64_t)(((uint32_t)axis << 30) | (uint32_t)right) << 32);
        build(a + nl, tmp + nl, len - nl, &sr, level + 1, out, n_out, leaf_out, n_leaf, loc);
    }
};

struct Piece {  // a node of the level-parallel top of the tree, or a deferred sub-tree task
    bool is_task = false;
    bool is_leaf = false;
    Rec leaf_rec{};
    __m128 sum = _mm_setzero_ps();  // Σ of the range's points, left by the parent's partition (the root sums its own)
    bool has_sum = false;
    int axis = 0;
    float th = 0.f;
    int left = -1, right = -1;  // piece indices
    size_t off = 0, len = 0;    // range in idx
    int level = 0;
    std::vector<uint64_t> slots;  // task output
    size_t n_slots = 0, n_leaves = 0;  // size of this piece's sub-tree output (filled bottom-up)
    size_t pos = 0, leaf_pos = 0;      // where it starts in the packed tree / in the leaf list
    bool bounded = true;
};

// A small persistent pool: target ingest is called once per keyframe by the streaming front-end (≈35 k points), where spawning
// threads per call would cost more than the build. run(n, fn) calls fn(i) for i in [0, n) on the pool and the caller's thread.
// Threads the pool may have: the host's, capped at 48, divided by the ranks that share the host when the process was started by
// a launcher (LOCAL_WORLD_SIZE / WORLD_SIZE: eight ranks must not spin 8 × 47 idle threads), or LOCGPU_BUILD_THREADS.
inline unsigned pool_thread_budget() {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 48) nt = 48;
    const char* lw = std::getenv("LOCAL_WORLD_SIZE");
    if (!lw) lw = std::getenv("WORLD_SIZE");
    const int ranks = lw ? std::atoi(lw) : 1;
    if (ranks > 1) nt = std::max(2u, nt / (unsigned)ranks);
    if (const char* e = std::getenv("LOCGPU_BUILD_THREADS")) { const int v = std::atoi(e); if (v >= 1) nt = (unsigned)std::min(v, 256); }
    return nt;
}

class Pool {
public:
    // A forked child (Python multiprocessing after an ingest in the parent) inherits the object but none of its threads: a pool
    // that belongs to another process is abandoned — its mutexes may have been held at the fork — and a fresh one is built.
    static Pool& get() {
        static std::mutex mu;
        static Pool* p = nullptr;
        std::lock_guard<std::mutex> lk(mu);
        if (!p || p->pid_ != ::getpid()) p = new Pool();
        return *p;
    }
    unsigned size() const { return (unsigned)workers_.size() + 1; }
    // fn(i) for i in [0, n) on the caller's thread and at most max_threads − 1 pool threads. Only as many workers as can be useful
    // are woken: waking (and waiting for) every thread of a 256-thread host costs more than a 35 k-point build.
    template <class F> void run(size_t n, unsigned max_threads, F&& fn) {
        if (n == 0) return;
        const size_t helpers = std::min<size_t>({workers_.size(), n - 1, max_threads > 0 ? (size_t)max_threads - 1 : 0});
        if (helpers == 0) { for (size_t i = 0; i < n; ++i) fn(i); return; }
        std::lock_guard<std::mutex> serial(run_mu_);  // one parallel region at a time (several contexts may ingest concurrently)
        std::function<void(size_t)> f = fn;
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &f; n_ = n; next_.store(0); pending_ = helpers; tickets_ = helpers;
        }
        for (size_t h = 0; h < helpers; ++h) cv_.notify_one();
        for (size_t i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) f(i);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    Pool() : pid_(::getpid()) {
        const unsigned nt = pool_thread_budget();
        for (unsigned t = 1; t < nt; ++t) workers_.emplace_back([this] { loop(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void loop() {
        for (;;) {
            std::function<void(size_t)>* f;
            size_t n;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return tickets_ > 0 || stop_; });
                if (stop_) return;
                --tickets_;
                f = fn_; n = n_;
            }
            for (size_t i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*f)(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_cv_.notify_one();
        }
    }
    const pid_t pid_;
    std::vector<std::thread> workers_;
    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_cv_;
    std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0, pending_ = 0, tickets_ = 0;
    std::atomic<size_t> next_{0};
    bool stop_ = false;
};

inline bool slot_floats_bounded(const uint64_t* s, size_t n) {
    for (size_t i = 0; i < n;) {
        const uint32_t meta = (uint32_t)(s[i] >> 32);
        const bool leaf = (meta >> 30) == 3u;
        float f[3];
        const uint32_t w0 = (uint32_t)s[i];
        std::memcpy(&f[0], &w0, 4);
        int nf = 1;
        if (leaf) {
            const uint32_t w1 = (uint32_t)s[i + 1], w2 = (uint32_t)(s[i + 1] >> 32);
            std::memcpy(&f[1], &w1, 4); std::memcpy(&f[2], &w2, 4);
            nf = 3;
        }
        for (int k = 0; k < nf; ++k)
            if (!(std::fabs(f[k]) < 1e18f)) return false;
        i += leaf ? 2 : 1;
    }
    return true;
}

}  // namespace

// Scratch of a build — the point records and their partition buffer — kept between builds: a keyframe front-end re-ingests a
// ≈35 k-point local map every few scans, and two fresh 0.5 MB vectors per build are two mmap/munmap pairs and ≈300 page faults
// (a fifth of such a build). A small free list (several contexts may build at once); capacity only grows.
struct BuildScratch { std::vector<Rec> a, tmp; };
class ScratchCache {
public:
    static std::unique_ptr<BuildScratch> acquire() {
        std::lock_guard<std::mutex> lk(mu());
        auto& f = free_list();
        if (f.empty()) return std::make_unique<BuildScratch>();
        std::unique_ptr<BuildScratch> s = std::move(f.back());
        f.pop_back();
        return s;
    }
    static void release(std::unique_ptr<BuildScratch> s) {
        std::lock_guard<std::mutex> lk(mu());
        auto& f = free_list();
        if (f.size() < 4 && s->a.size() <= (size_t)4 << 20) f.push_back(std::move(s));  // keep at most four, none larger than 64 MB a vector (a 10 M-point map's goes back)
    }
private:
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::vector<std::unique_ptr<BuildScratch>>& free_list() { static std::vector<std::unique_ptr<BuildScratch>> f; return f; }
};

// The build every map without duplicate points takes. A sub-tree of m distinct points is m leaves (two slots each) and m − 1 nodes:
// 3m − 1 slots, so in preorder a node at slot s with nl points on its left has its left child at s + 1 and its right child at
// s + 3·nl — known the moment the node is split. Nothing has to wait for a level to finish or for sizes to come back: a thread
// splits a node, writes its slot, hands the right child to a shared list and goes on with the left one; ranges of at most
// task_len points are finished recursively, straight into the final arrays. The threads of the pool are woken ONCE per build
// (behind the root's split, which nobody can help with) and look for work until none is outstanding — the level-by-level
// scheme below paid a condition-variable round trip per level, a third of a 35 k-point build.
// Returns false when a degenerate split (every point of a node on one side — duplicates, kdtree.cpp:66-70) made a sub-tree
// smaller than that: the caller restores the records and takes the level-by-level path, which measures sizes before it places.
struct WorkItem { size_t off, len, pos, leaf_pos; int level; __m128 sum; bool has_sum; };

bool build_direct(Builder& b, Rec* a, Rec* tmp, size_t n, Pool& pool, unsigned nt, size_t task_len, PackedKdTree& out, bool& all_bounded) {
    out.slots.resize(3 * n - 1);
    out.leaf_slots.resize(n);
    uint64_t* slots = out.slots.data();
    uint32_t* leaves = out.leaf_slots.data();
    std::mutex qmu;
    std::condition_variable qcv;
    std::vector<WorkItem> queue;
    queue.reserve(1024);
    long outstanding = 1;               // items in the list or being worked on; guarded by qmu
    std::atomic<long> listed{1};        // queue.size(), readable without the lock
    std::atomic<bool> finished{false};  // outstanding reached 0
    std::atomic<bool> ok{true}, bounded{true};
    queue.push_back(WorkItem{0, n, 0, 0, 1, _mm_setzero_ps(), false});
    auto take = [&](WorkItem& w) {  // under qmu: the largest range first — it has the longest way to go
        if (queue.empty()) return false;
        size_t best = 0;
        for (size_t i = 1; i < queue.size(); ++i)
            if (queue[i].len > queue[best].len) best = i;
        w = queue[best];
        queue[best] = queue.back();
        queue.pop_back();
        listed.store((long)queue.size(), std::memory_order_relaxed);
        return true;
    };
    pool.run(nt, nt, [&](size_t) {
        Local loc;
        for (;;) {
            WorkItem w{};
            bool got = false;
            // An idle thread polls for a moment — work turns up every few µs while the top of the tree is being split, and a futex
            // wake-up (tens of µs each, one after the other down the right spine) is what the level-by-level scheme lost its time
            // to — and then SLEEPS: a helper spinning on a shared or over-subscribed core slows the thread that has the work.
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 1;; ++it) {
                if (listed.load(std::memory_order_acquire) > 0) {
                    std::lock_guard<std::mutex> lk(qmu);
                    got = take(w);
                }
                if (got || finished.load(std::memory_order_acquire) || !ok.load(std::memory_order_relaxed)) break;
                _mm_pause();
                if ((it & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) break;
            }
            if (!got && !finished.load(std::memory_order_acquire) && ok.load(std::memory_order_relaxed)) {
                std::unique_lock<std::mutex> lk(qmu);
                qcv.wait(lk, [&] { return !queue.empty() || outstanding == 0 || !ok.load(std::memory_order_relaxed); });
                if (ok.load(std::memory_order_relaxed)) got = take(w);
            }
            if (!got) break;
            while (ok.load(std::memory_order_relaxed)) {  // down the left spine
                if (w.len <= task_len) {
                    size_t n_out = w.pos, n_leaf = w.leaf_pos;
                    b.build(a + w.off, tmp + w.off, w.len, w.has_sum ? &w.sum : nullptr, w.level, slots, n_out, leaves, n_leaf, loc);
                    if (n_out - w.pos != 3 * w.len - 1) ok.store(false);
                    else if (!slot_floats_bounded(slots + w.pos, 3 * w.len - 1)) bounded.store(false);
                    break;
                }
                if (w.level > loc.depth) loc.depth = w.level;
                int axis; float th; size_t nl;
                __m128 sl, sr;
                if (!b.split(a + w.off, tmp + w.off, w.len, w.has_sum ? &w.sum : nullptr, axis, th, nl, sl, sr)) { ok.store(false); break; }
                const size_t right_pos = w.pos + 3 * nl;
                uint32_t tb; std::memcpy(&tb, &th, 4);
                slots[w.pos] = (uint64_t)tb | ((uint64_t)(((uint32_t)axis << 30) | (uint32_t)right_pos) << 32);
                if (!(std::fabs(th) < 1e18f)) bounded.store(false);
                {
                    std::lock_guard<std::mutex> lk(qmu);
                    ++outstanding;
                    queue.push_back(WorkItem{w.off + nl, w.len - nl, right_pos, w.leaf_pos + nl, w.level + 1, sr, true});
                    listed.store((long)queue.size(), std::memory_order_release);
                }
                qcv.notify_one();
                w = WorkItem{w.off, nl, w.pos + 1, w.leaf_pos, w.level + 1, sl, true};
            }
            bool last;
            {
                std::lock_guard<std::mutex> lk(qmu);
                last = --outstanding == 0;
                if (last) finished.store(true, std::memory_order_release);
            }
            if (last || !ok.load(std::memory_order_relaxed)) qcv.notify_all();
        }
        b.merge(loc);
    });
    all_bounded = bounded.load();
    return ok.load();
}

bool build_packed_kdtree(const float* xyz, size_t n, PackedKdTree& out, std::string& err) {
    out.slots.clear(); out.leaf_slots.clear();  // a PackedKdTree that is reused keeps its capacity (same reason as BuildScratch)
    out.num_leaves = out.num_nodes = out.num_points = 0; out.depth = 0; out.bounded = true;
    if (n == 0) { err = "empty target cloud"; return false; }
    // 3n-1 slots of 8 bytes must stay below 4 GiB: the search kernel addresses the tree through a 32-bit buffer offset
    if (n >= (1ull << 29) / 3) { err = "target cloud too large (the packed tree must stay below 4 GiB)"; return false; }
    static const bool times = std::getenv("LOCGPU_BUILD_TIMES") != nullptr;  // diagnostic: phase times of every build on stderr
    auto t_prev = std::chrono::steady_clock::now();
    double t_phase[5] = {0, 0, 0, 0, 0};  // records, root split, further top levels, tasks, lay-out + top slots
    auto lap = [&](int ph) {
        if (!times) return;
        const auto t = std::chrono::steady_clock::now();
        t_phase[ph] += std::chrono::duration<double, std::micro>(t - t_prev).count();
        t_prev = t;
    };
    Builder b;
    struct ScratchHold { std::unique_ptr<BuildScratch> s = ScratchCache::acquire(); ~ScratchHold() { ScratchCache::release(std::move(s)); } } hold;
    std::vector<Rec>& idx = hold.s->a;   // the points with their indices, reordered node by node
    std::vector<Rec>& tmp = hold.s->tmp;  // partition buffer of the same size
    if (idx.size() < n) { idx.resize(n); tmp.resize(n); }
    for (size_t i = 0; i < n; ++i) idx[i] = Rec{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], (int32_t)i};
    lap(0);

    Pool& pool = Pool::get();
    static const size_t grain = [] { const char* e = std::getenv("LOCGPU_BUILD_GRAIN"); const long v = e ? std::atol(e) : 0; return v >= 256 ? (size_t)v : (size_t)2048; }();
    const unsigned nt = (unsigned)std::min<size_t>(pool.size(), std::max<size_t>(1, n / grain));  // threads worth waking for this map
    {
        bool all_bounded = true;
        static const size_t min_task = [] { const char* e = std::getenv("LOCGPU_BUILD_TASK"); const long v = e ? std::atol(e) : 0; return v >= 16 ? (size_t)v : (size_t)1024; }();
        if (build_direct(b, idx.data(), tmp.data(), n, pool, nt, std::max<size_t>(n / (8 * (size_t)nt), min_task), out, all_bounded)) {
            lap(3);
            if (times) std::fprintf(stderr, "[locgpu build] %zu points, %u threads: records %.0f us | direct build %.0f us\n", n, nt, t_phase[0], t_phase[3]);
            out.num_leaves = (size_t)b.leaves.load();
            out.num_nodes = out.slots.size() - out.num_leaves;
            out.depth = b.depth.load();
            out.num_points = n;
            out.bounded = all_bounded;
            return true;
        }
        // duplicates: start again from the records' original order, sizes first
        b.leaves.store(0);
        b.depth.store(0);
        for (size_t i = 0; i < n; ++i) idx[i] = Rec{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], (int32_t)i};
        lap(0);
    }
    const size_t task_len = std::max<size_t>(n / (4 * (size_t)nt), 2048);

    // Top of the tree, level by level: the nodes of one level are independent, so they are split in parallel (each node still
    // sums its points sequentially in index order — math_utils.h:40-45); sub-trees of at most task_len points become tasks.
    std::vector<Piece> pieces(1);
    pieces.reserve(16 * (n / task_len + 2));
    pieces[0].off = 0; pieces[0].len = n; pieces[0].level = 1;
    std::vector<int> frontier{0}, tasks;
    while (!frontier.empty()) {
        std::vector<int> split_now;
        for (int pi : frontier) {
            if (pieces[pi].len <= task_len) { pieces[pi].is_task = true; tasks.push_back(pi); }
            else split_now.push_back(pi);
        }
        const size_t base = pieces.size();
        pieces.resize(base + 2 * split_now.size());  // two child slots per node, reserved before the parallel region
        pool.run(split_now.size(), nt, [&](size_t k) {
            Piece& p = pieces[split_now[k]];
            b.note_depth(p.level);
            const Rec first = idx[p.off];
            int axis; float th; size_t nl;
            __m128 sl, sr;
            if (!b.split(idx.data() + p.off, tmp.data() + p.off, p.len, p.has_sum ? &p.sum : nullptr, axis, th, nl, sl, sr)) {
                p.is_leaf = true; p.leaf_rec = first;
                return;
            }
            p.axis = axis; p.th = th;
            p.left = (int)(base + 2 * k); p.right = p.left + 1;
            Piece& l = pieces[p.left];
            Piece& r = pieces[p.right];
            l.off = p.off; l.len = nl; l.level = p.level + 1; l.sum = sl; l.has_sum = true;
            r.off = p.off + nl; r.len = p.len - nl; r.level = p.level + 1; r.sum = sr; r.has_sum = true;
        });
        frontier.clear();
        for (int pi : split_now)
            if (!pieces[pi].is_leaf) { frontier.push_back(pieces[pi].left); frontier.push_back(pieces[pi].right); }
        lap(split_now.size() == 1 && split_now[0] == 0 ? 1 : 2);
    }

    // The tasks, largest first. A sub-tree of m distinct points is m leaves (two slots each) and m − 1 nodes: 3m − 1 slots. Only a
    // degenerate split (every point of a node on one side — duplicates, kdtree.cpp:66-70) makes it smaller, so positions are laid out
    // for the full size first and every task writes its sub-tree straight into the final arrays; if one of them comes out
    // smaller the tasks are run again — from their ranges' original order — into buffers of their own and assembled (the path
    // every build took before).
    std::sort(tasks.begin(), tasks.end(), [&](int a, int c) { return pieces[a].len > pieces[c].len; });
    auto lay_out = [&] {  // sizes bottom-up (children have larger indices than their parent), then positions in preorder
        for (size_t i = pieces.size(); i-- > 0;) {
            Piece& p = pieces[i];
            if (p.is_task) continue;
            if (p.is_leaf) { p.n_slots = 2; p.n_leaves = 1; }
            else if (p.left >= 0) { p.n_slots = 1 + pieces[p.left].n_slots + pieces[p.right].n_slots; p.n_leaves = pieces[p.left].n_leaves + pieces[p.right].n_leaves; }
        }
        std::vector<int> stack{0};
        pieces[0].pos = 0; pieces[0].leaf_pos = 0;
        while (!stack.empty()) {
            const Piece& p = pieces[stack.back()];
            stack.pop_back();
            if (p.is_task || p.is_leaf) continue;
            Piece& l = pieces[p.left];
            Piece& r = pieces[p.right];
            l.pos = p.pos + 1; l.leaf_pos = p.leaf_pos;
            r.pos = l.pos + l.n_slots; r.leaf_pos = l.leaf_pos + l.n_leaves;
            stack.push_back(p.right);
            stack.push_back(p.left);
        }
    };
    for (int ti : tasks) { pieces[ti].n_slots = 3 * pieces[ti].len - 1; pieces[ti].n_leaves = pieces[ti].len; }
    lay_out();
    out.slots.resize(pieces[0].n_slots);
    out.leaf_slots.resize(pieces[0].n_leaves);
    std::atomic<bool> direct_ok{true};
    pool.run(tasks.size(), nt, [&](size_t t) {
        Piece& p = pieces[tasks[t]];
        Local loc;
        size_t n_out = p.pos, n_leaf = p.leaf_pos;
        b.build(idx.data() + p.off, tmp.data() + p.off, p.len, p.has_sum ? &p.sum : nullptr, p.level, out.slots.data(), n_out, out.leaf_slots.data(), n_leaf, loc);
        if (n_out - p.pos != p.n_slots) { direct_ok.store(false); return; }  // a degenerate split inside: the layout was too generous
        b.merge(loc);
        p.bounded = slot_floats_bounded(out.slots.data() + p.pos, p.n_slots);
    });
    const bool direct = direct_ok.load();
    lap(3);
    if (!direct) {
        b.leaves.store(0);
        pool.run(tasks.size(), nt, [&](size_t t) {
            Piece& p = pieces[tasks[t]];
            p.slots.resize(3 * p.len);
            // back to the order the first pass found: every node's range is in increasing point index (the root's is, and a stable
            // partition keeps it), and the variance sums — which pick the axis — must run over it in exactly that order
            std::sort(idx.begin() + p.off, idx.begin() + p.off + p.len, [](const Rec& x, const Rec& y) { return x.id < y.id; });
            Local loc;
            size_t n_out = 0, n_leaf = 0;
            b.build(idx.data() + p.off, tmp.data() + p.off, p.len, p.has_sum ? &p.sum : nullptr, p.level, p.slots.data(), n_out, nullptr, n_leaf, loc);
            p.slots.resize(n_out);
            b.merge(loc);
            p.n_slots = p.slots.size();
            p.n_leaves = (size_t)loc.leaves;
            p.bounded = slot_floats_bounded(p.slots.data(), p.slots.size());
        });
        lay_out();
    }
    const size_t total = pieces[0].n_slots;
    out.slots.resize(total);
    out.leaf_slots.resize(pieces[0].n_leaves);
    std::atomic<bool> bounded{true};
    if (direct) {  // only the top of the tree is left to write: single slots, a few dozen of them
        for (Piece& p : pieces) {
            if (p.is_task) { if (!p.bounded) bounded.store(false); continue; }
            if (p.is_leaf) {
                Local loc;
                size_t at = p.pos;
                Builder::emit_leaf(out.slots.data(), at, p.leaf_rec, loc);
                b.merge(loc);
                out.leaf_slots[p.leaf_pos] = (uint32_t)p.pos;
                if (!slot_floats_bounded(out.slots.data() + p.pos, 2)) bounded.store(false);
            } else if (p.left >= 0) {
                uint32_t tb; std::memcpy(&tb, &p.th, 4);
                out.slots[p.pos] = (uint64_t)tb | ((uint64_t)(((uint32_t)p.axis << 30) | (uint32_t)pieces[p.right].pos) << 32);
                if (!(std::fabs(p.th) < 1e18f)) bounded.store(false);
            }
        }
    } else
    // Assemble: task outputs are copied (right-child indices rebased by the task's position) in parallel; top nodes are single slots.
    pool.run(pieces.size(), nt, [&](size_t i) {
        Piece& p = pieces[i];
        if (p.is_task) {
            const size_t base = p.pos;
            size_t lp = p.leaf_pos;
            for (size_t k = 0; k < p.slots.size(); ++k) {
                uint64_t sl = p.slots[k];
                const uint32_t meta = (uint32_t)(sl >> 32);
                if ((meta >> 30) != 3u) {
                    const uint32_t rebased = (meta & 0x3FFFFFFFu) + (uint32_t)base;
                    out.slots[base + k] = (sl & 0xFFFFFFFFull) | ((uint64_t)((meta & 0xC0000000u) | rebased) << 32);
                } else {
                    out.slots[base + k] = sl;
                    out.leaf_slots[lp++] = (uint32_t)(base + k);
                    out.slots[base + k + 1] = p.slots[k + 1];  // second leaf slot {y,z}: raw floats, never rebased
                    ++k;
                }
            }
            if (!p.bounded) bounded.store(false);
            std::vector<uint64_t>().swap(p.slots);
        } else if (p.is_leaf) {
            uint64_t two[2];
            Local loc;
            size_t n_two = 0;
            Builder::emit_leaf(two, n_two, p.leaf_rec, loc);
            b.merge(loc);
            out.slots[p.pos] = two[0];
            out.slots[p.pos + 1] = two[1];
            out.leaf_slots[p.leaf_pos] = (uint32_t)p.pos;
            if (!slot_floats_bounded(two, 2)) bounded.store(false);
        } else if (p.left >= 0) {
            uint32_t tb; std::memcpy(&tb, &p.th, 4);
            out.slots[p.pos] = (uint64_t)tb | ((uint64_t)(((uint32_t)p.axis << 30) | (uint32_t)pieces[p.right].pos) << 32);
            if (!(std::fabs(p.th) < 1e18f)) bounded.store(false);
        }
    });
    lap(4);
    if (times)
        std::fprintf(stderr, "[locgpu build] %zu points, %u threads, %zu tasks: records %.0f us | root split %.0f | further top levels %.0f | tasks %.0f | lay-out + top slots %.0f\n", n, nt,
                     tasks.size(), t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4]);
    out.num_leaves = (size_t)b.leaves.load();
    out.num_nodes = total - out.num_leaves;  // internal (1 slot) + leaves (2 slots) ⇒ nodes = slots − leaves
    out.depth = b.depth.load();
    out.num_points = n;
    // The fast search kernel assumes squared distances cannot overflow or be NaN: true when every float in the tree (leaf
    // coordinates and split thresholds) is finite and small enough. Otherwise searches use the exact kernel only.
    out.bounded = bounded.load();
    return true;
}

}  // namespace locgpu

// Host-only test hook (not part of include/locgpu.h): builds the packed tree and copies it out so the CPU test-suite
// can compare its structure with the oracle's tree node by node. Returns the number of slots (or 0 on failure);
// info = {leaves, nodes, depth}.
extern "C" __attribute__((visibility("default"))) size_t locgpu_debug_build_tree(const float* xyz, size_t n, uint64_t* slots, size_t cap,
                                                                                  int64_t info[3]) {
    locgpu::PackedKdTree t;
    std::string err;
    if (!locgpu::build_packed_kdtree(xyz, n, t, err)) return 0;
    if (info) { info[0] = (int64_t)t.num_leaves; info[1] = (int64_t)t.num_nodes; info[2] = t.depth; }
    if (slots) std::memcpy(slots, t.slots.data(), std::min(cap, t.slots.size()) * sizeof(uint64_t));
    return t.slots.size();
}
