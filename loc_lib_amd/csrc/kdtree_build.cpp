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

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace locgpu {
namespace {

struct Local {  // per-task counters (shared atomics on every leaf cost more than the build itself on small maps)
    int64_t leaves = 0;
    int depth = 0;
};

struct Builder {
    const float* pts;  // packed xyz, 3 floats per point
    std::atomic<int64_t> leaves{0};
    std::atomic<int> depth{0};
    void merge(const Local& l) {
        leaves.fetch_add(l.leaves, std::memory_order_relaxed);
        int d = depth.load(std::memory_order_relaxed);
        while (l.depth > d && !depth.compare_exchange_weak(d, l.depth, std::memory_order_relaxed)) {}
    }

    // Split [idx, idx+len) the way FindSplitAxisAndThresh does. Returns false for the degenerate case.
    bool split(int32_t* idx, int32_t* tmp, size_t len, int& axis, float& th, size_t& n_left) const {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (size_t i = 0; i < len; ++i) {
            const float* p = pts + 3 * (size_t)idx[i];
            sx = sx + p[0]; sy = sy + p[1]; sz = sz + p[2];
        }
        const float flen = (float)len;
        const float mx = sx / flen, my = sy / flen, mz = sz / flen;
        float vx = 0.f, vy = 0.f, vz = 0.f;
        for (size_t i = 0; i < len; ++i) {
            const float* p = pts + 3 * (size_t)idx[i];
            const float dx = p[0] - mx, dy = p[1] - my, dz = p[2] - mz;
            vx = vx + dx * dx; vy = vy + dy * dy; vz = vz + dz * dz;
        }
        const float flen1 = (float)(len - 1);
        vx = vx / flen1; vy = vy / flen1; vz = vz / flen1;
        axis = 0;
        float best = vx;
        if (vy > best) { best = vy; axis = 1; }
        if (vz > best) { best = vz; axis = 2; }
        th = axis == 0 ? mx : (axis == 1 ? my : mz);
        size_t nl = 0, nr = 0;
        for (size_t i = 0; i < len; ++i) {  // stable partition: `< th` left, else right
            const int32_t id = idx[i];
            if (pts[3 * (size_t)id + axis] < th) idx[nl++] = id;
            else tmp[nr++] = id;
        }
        std::memcpy(idx + nl, tmp, nr * sizeof(int32_t));
        n_left = nl;
        return !(nl == 0 || nr == 0);
    }

    void emit_leaf(std::vector<uint64_t>& out, int32_t id, Local& loc) {
        const float* p = pts + 3 * (size_t)id;
        uint32_t xb, yb, zb;
        std::memcpy(&xb, p, 4); std::memcpy(&yb, p + 1, 4); std::memcpy(&zb, p + 2, 4);
        out.push_back((uint64_t)xb | ((uint64_t)((3u << 30) | (uint32_t)id) << 32));
        out.push_back((uint64_t)yb | ((uint64_t)zb << 32));
        loc.leaves++;
    }

    void note_depth(int level) {
        int d = depth.load(std::memory_order_relaxed);
        while (level > d && !depth.compare_exchange_weak(d, level, std::memory_order_relaxed)) {}
    }

    // Recursive build of one sub-tree into `out` (slot indices relative to out's start).
    void build(int32_t* idx, int32_t* tmp, size_t len, int level, std::vector<uint64_t>& out, Local& loc) {
        if (level > loc.depth) loc.depth = level;
        if (len == 1) { emit_leaf(out, idx[0], loc); return; }
        const int32_t first = idx[0];  // points[0] before the partition reorders nothing: stable ⇒ idx[0] stays first of its side
        int axis; float th; size_t nl;
        if (!split(idx, tmp, len, axis, th, nl)) { emit_leaf(out, first, loc); return; }
        const size_t pos = out.size();
        out.push_back(0);
        build(idx, tmp, nl, level + 1, out, loc);
        const size_t right = out.size();
        uint32_t tb; std::memcpy(&tb, &th, 4);
        out[pos] = (uint64_t)tb | ((uint64_t)(((uint32_t)axis << 30) | (uint32_t)right) << 32);
        build(idx + nl, tmp + nl, len - nl, level + 1, out, loc);
    }
};

struct Piece {  // a node of the level-parallel top of the tree, or a deferred sub-tree task
    bool is_task = false;
    bool is_leaf = false;
    int32_t leaf_id = 0;
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

bool build_packed_kdtree(const float* xyz, size_t n, PackedKdTree& out, std::string& err) {
    out = PackedKdTree();
    if (n == 0) { err = "empty target cloud"; return false; }
    // 3n-1 slots of 8 bytes must stay below 4 GiB: the search kernel addresses the tree through a 32-bit buffer offset
    if (n >= (1ull << 29) / 3) { err = "target cloud too large (the packed tree must stay below 4 GiB)"; return false; }
    Builder b;
    b.pts = xyz;
    std::vector<int32_t> idx(n), tmp(n);
    for (size_t i = 0; i < n; ++i) idx[i] = (int32_t)i;

    Pool& pool = Pool::get();
    const unsigned nt = (unsigned)std::min<size_t>(pool.size(), std::max<size_t>(1, n / 4096));  // threads worth waking for this map
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
            const int32_t first = idx[p.off];
            int axis; float th; size_t nl;
            if (!b.split(idx.data() + p.off, tmp.data() + p.off, p.len, axis, th, nl)) {
                p.is_leaf = true; p.leaf_id = first;
                return;
            }
            p.axis = axis; p.th = th;
            p.left = (int)(base + 2 * k); p.right = p.left + 1;
            Piece& l = pieces[p.left];
            Piece& r = pieces[p.right];
            l.off = p.off; l.len = nl; l.level = p.level + 1;
            r.off = p.off + nl; r.len = p.len - nl; r.level = p.level + 1;
        });
        frontier.clear();
        for (int pi : split_now)
            if (!pieces[pi].is_leaf) { frontier.push_back(pieces[pi].left); frontier.push_back(pieces[pi].right); }
    }

    // The tasks, largest first.
    std::sort(tasks.begin(), tasks.end(), [&](int a, int c) { return pieces[a].len > pieces[c].len; });
    pool.run(tasks.size(), nt, [&](size_t t) {
        Piece& p = pieces[tasks[t]];
        p.slots.reserve(3 * p.len);
        Local loc;
        b.build(idx.data() + p.off, tmp.data() + p.off, p.len, p.level, p.slots, loc);
        b.merge(loc);
        p.n_slots = p.slots.size();
        p.n_leaves = (size_t)loc.leaves;
        p.bounded = slot_floats_bounded(p.slots.data(), p.slots.size());
    });

    // Sizes bottom-up (children have larger indices than their parent), then positions in preorder.
    for (size_t i = pieces.size(); i-- > 0;) {
        Piece& p = pieces[i];
        if (p.is_task) continue;
        if (p.is_leaf) { p.n_slots = 2; p.n_leaves = 1; }
        else if (p.left >= 0) { p.n_slots = 1 + pieces[p.left].n_slots + pieces[p.right].n_slots; p.n_leaves = pieces[p.left].n_leaves + pieces[p.right].n_leaves; }
    }
    {
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
    }
    const size_t total = pieces[0].n_slots;
    out.slots.resize(total);
    out.leaf_slots.resize(pieces[0].n_leaves);
    std::atomic<bool> bounded{true};
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
            std::vector<uint64_t> two;
            Local loc;
            b.emit_leaf(two, p.leaf_id, loc);
            b.merge(loc);
            out.slots[p.pos] = two[0];
            out.slots[p.pos + 1] = two[1];
            out.leaf_slots[p.leaf_pos] = (uint32_t)p.pos;
            if (!slot_floats_bounded(two.data(), 2)) bounded.store(false);
        } else if (p.left >= 0) {
            uint32_t tb; std::memcpy(&tb, &p.th, 4);
            out.slots[p.pos] = (uint64_t)tb | ((uint64_t)(((uint32_t)p.axis << 30) | (uint32_t)pieces[p.right].pos) << 32);
            if (!(std::fabs(p.th) < 1e18f)) bounded.store(false);
        }
    });
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
