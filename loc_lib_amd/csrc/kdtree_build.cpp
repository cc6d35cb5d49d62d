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

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace locgpu {
namespace {

struct Builder {
    const float* pts;  // packed xyz, 3 floats per point
    std::atomic<int64_t> leaves{0};
    std::atomic<int> depth{0};

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

    void emit_leaf(std::vector<uint64_t>& out, int32_t id) {
        const float* p = pts + 3 * (size_t)id;
        uint32_t xb, yb, zb;
        std::memcpy(&xb, p, 4); std::memcpy(&yb, p + 1, 4); std::memcpy(&zb, p + 2, 4);
        out.push_back((uint64_t)xb | ((uint64_t)((3u << 30) | (uint32_t)id) << 32));
        out.push_back((uint64_t)yb | ((uint64_t)zb << 32));
        leaves.fetch_add(1, std::memory_order_relaxed);
    }

    void note_depth(int level) {
        int d = depth.load(std::memory_order_relaxed);
        while (level > d && !depth.compare_exchange_weak(d, level, std::memory_order_relaxed)) {}
    }

    // Recursive build of one sub-tree into `out` (slot indices relative to out's start).
    void build(int32_t* idx, int32_t* tmp, size_t len, int level, std::vector<uint64_t>& out) {
        note_depth(level);
        if (len == 1) { emit_leaf(out, idx[0]); return; }
        const int32_t first = idx[0];  // points[0] before the partition reorders nothing: stable ⇒ idx[0] stays first of its side
        int axis; float th; size_t nl;
        if (!split(idx, tmp, len, axis, th, nl)) { emit_leaf(out, first); return; }
        const size_t pos = out.size();
        out.push_back(0);
        build(idx, tmp, nl, level + 1, out);
        const size_t right = out.size();
        uint32_t tb; std::memcpy(&tb, &th, 4);
        out[pos] = (uint64_t)tb | ((uint64_t)(((uint32_t)axis << 30) | (uint32_t)right) << 32);
        build(idx + nl, tmp + nl, len - nl, level + 1, out);
    }
};

struct Piece {  // a node of the serially built top of the tree, or a deferred sub-tree task
    bool is_task = false;
    bool is_leaf = false;
    int32_t leaf_id = 0;
    int axis = 0;
    float th = 0.f;
    int left = -1, right = -1;  // piece indices
    size_t off = 0, len = 0;    // task: range in idx
    int level = 0;
    std::vector<uint64_t> slots;  // task output
};

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

    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 4;
    if (nt > 64) nt = 64;
    const size_t task_len = std::max<size_t>(n / (8 * (size_t)nt), 4096);

    // Top of the tree, serially (each node still sums in index order); sub-trees below task_len become tasks.
    std::vector<Piece> pieces;
    pieces.reserve(64 * nt + 64);
    struct Frame { int piece; size_t off, len; int level; };
    std::vector<Frame> todo;
    pieces.emplace_back();
    todo.push_back({0, 0, n, 1});
    while (!todo.empty()) {
        const Frame f = todo.back();
        todo.pop_back();
        Piece& p = pieces[f.piece];
        p.level = f.level;
        if (f.len <= task_len) { p.is_task = true; p.off = f.off; p.len = f.len; continue; }
        b.note_depth(f.level);
        const int32_t first = idx[f.off];
        int axis; float th; size_t nl;
        if (!b.split(idx.data() + f.off, tmp.data() + f.off, f.len, axis, th, nl)) {
            p.is_leaf = true; p.leaf_id = first;
            continue;
        }
        p.axis = axis; p.th = th;
        const int li = (int)pieces.size();
        pieces.emplace_back();
        pieces.emplace_back();
        pieces[f.piece].left = li;  // (re-index: emplace_back may have moved `p`)
        pieces[f.piece].right = li + 1;
        todo.push_back({li + 1, f.off + nl, f.len - nl, f.level + 1});
        todo.push_back({li, f.off, nl, f.level + 1});
    }

    // Run the tasks.
    std::vector<int> tasks;
    for (size_t i = 0; i < pieces.size(); ++i) if (pieces[i].is_task) tasks.push_back((int)i);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= tasks.size()) break;
            Piece& p = pieces[tasks[t]];
            p.slots.reserve(3 * p.len);
            b.build(idx.data() + p.off, tmp.data() + p.off, p.len, p.level, p.slots);
        }
    };
    std::vector<std::thread> th;
    const unsigned nthreads = (unsigned)std::min<size_t>(nt, tasks.size());
    for (unsigned t = 1; t < nthreads; ++t) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();

    // Assemble in preorder; right-child indices inside task outputs are rebased by the task's base slot.
    size_t total = 0;
    for (const Piece& p : pieces) total += p.is_task ? p.slots.size() : (p.is_leaf ? 2 : 1);
    out.slots.resize(total);
    struct AFrame { int piece; };
    std::vector<int> stack{0};
    size_t pos = 0;
    std::vector<std::pair<size_t, int>> pending_right;  // (slot position of internal node, right piece) resolved when reached
    std::vector<size_t> piece_pos(pieces.size(), 0);
    while (!stack.empty()) {
        const int pi = stack.back();
        stack.pop_back();
        Piece& p = pieces[pi];
        piece_pos[pi] = pos;
        if (p.is_task) {
            const size_t base = pos;
            for (size_t i = 0; i < p.slots.size(); ++i) {
                uint64_t s = p.slots[i];
                const uint32_t meta = (uint32_t)(s >> 32);
                if ((meta >> 30) != 3u) {
                    const uint32_t rebased = (meta & 0x3FFFFFFFu) + (uint32_t)base;
                    s = (s & 0xFFFFFFFFull) | ((uint64_t)((meta & 0xC0000000u) | rebased) << 32);
                    out.slots[pos++] = s;
                } else {
                    out.slots[pos++] = s;
                    out.slots[pos++] = p.slots[++i];  // second leaf slot {y,z}: raw floats, never rebased
                }
            }
            std::vector<uint64_t>().swap(p.slots);
        } else if (p.is_leaf) {
            std::vector<uint64_t> two;
            b.emit_leaf(two, p.leaf_id);
            out.slots[pos++] = two[0];
            out.slots[pos++] = two[1];
        } else {
            pos++;  // patched once the right child's position is known
            stack.push_back(p.right);
            stack.push_back(p.left);
        }
    }
    for (size_t i = 0; i < pieces.size(); ++i) {
        const Piece& p = pieces[i];
        if (p.is_task || p.is_leaf) continue;
        uint32_t tb; std::memcpy(&tb, &p.th, 4);
        out.slots[piece_pos[i]] = (uint64_t)tb | ((uint64_t)(((uint32_t)p.axis << 30) | (uint32_t)piece_pos[p.right]) << 32);
    }
    out.num_leaves = (size_t)b.leaves.load();
    out.num_nodes = total - out.num_leaves;  // internal (1 slot) + leaves (2 slots) ⇒ nodes = slots − leaves
    out.depth = b.depth.load();
    out.num_points = n;
    // The fast search kernel assumes squared distances cannot overflow or be NaN: true when every float in the tree (leaf
    // coordinates and split thresholds) is finite and small enough. Otherwise searches use the exact kernel only.
    out.bounded = true;
    out.leaf_slots.reserve(out.num_leaves);
    for (size_t i = 0; i < out.slots.size();) {
        const uint32_t meta = (uint32_t)(out.slots[i] >> 32);
        const bool leaf = (meta >> 30) == 3u;
        if (leaf) out.leaf_slots.push_back((uint32_t)i);
        float f[3];
        const uint32_t w0 = (uint32_t)out.slots[i];
        std::memcpy(&f[0], &w0, 4);
        int nf = 1;
        if (leaf) {
            const uint32_t w1 = (uint32_t)out.slots[i + 1], w2 = (uint32_t)(out.slots[i + 1] >> 32);
            std::memcpy(&f[1], &w1, 4); std::memcpy(&f[2], &w2, 4);
            nf = 3;
        }
        for (int k = 0; k < nf; ++k)
            if (!(std::fabs(f[k]) < 1e18f)) out.bounded = false;
        i += leaf ? 2 : 1;
    }
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
