// loc_lib_amd/csrc/search_walk.hpp — the hot search traversal (K1), see DESIGN.md §3.
//
// The reference's recursion (KdTree::Knn / ComputeDisForLeaf / NeedExpand, kdtree.cpp:169-236) replayed node for node, organised so
// that a wave executes straight-line, fully predicated regions — no per-lane branches, hence no exec-mask bookkeeping on the CU's
// shared scalar unit:
//
//   * a lane that has nothing to visit (it is finished, or it sits out a stage) visits the SENTINEL LEAF behind the packed tree
//     (coordinates 3e38: its squared distance overflows to +inf and is never inserted), so "leaf" is the only predicate of a stage;
//   * the stack holds {far slot, −d²}. With the sign flipped, `d² < bound` is a SIGNED INTEGER compare of the raw bits against the
//     bits of −bound (non-positive floats order like their magnitudes), a row below the stack's bottom reads as 0 from outside the
//     workgroup's LDS allocation (never "less than" a non-positive bound) and INT_MIN (−0.0f) as the bound switches a test off:
//     the `j < avail` and `is this lane popping` predicates cost nothing;
//   * the push is an unconditional LDS store to the row above the top (garbage there is harmless; beyond the last row it falls off
//     the allocation) followed by `avail += pushed`;
//   * the first T pushes of a query are the top levels of its first descent (the result set is empty: NeedExpand is unconditionally
//     true). They are popped last, against the final bound, and almost never expanded — so they are not stored: the two smallest d²
//     among them are materialised as rows 0 and 1 of the stack after the first descent — older below younger, which is the
//     recursion's order — so the drain to them is an ordinary pop; only the third smallest d² stays in a register. A query for which
//     it could pass as well (≈1e-3 of them) is marked and recomputed by the DEEP PASS: the same traversal with every level stored,
//     over a device-side list. The other DF−2 rows hold one level each and T is chosen so that the first descent and everything
//     below a stored level fit; only the descent from a candidate, which starts on an un-stored level, can outgrow the rows — such a
//     query goes to the deep pass too;
//   * the result set is updated with v_med3_f32: inserting x into ascending d[0..K) and dropping the largest is
//     d'[j] = med3(d[j-1], x, d[j]), d'[0] = min(d[0], x).
//
// A query is flagged `slow` (→ exact recomputation with the libstdc++ heap restated move for move, KnnHeap) when an eviction happens
// while the maximum is tied or when two distances of its final set are equal.
#pragma once
#include "icp_kernels.hpp"

namespace locgpu {

constexpr float kSentinelCoord = 3.0e38f;  // the sentinel leaf behind the tree: (q − 3e38)² = +inf for every sane query

template <int K>
struct Walk {
    float qx, qy, qz;
    float d[K];        // ascending; +inf = empty
    uint32_t id[K];
    uint32_t cur;      // slot to visit next (`dummy` = the sentinel leaf: nothing to visit)
    int avail;         // rows on the stack
    uint32_t c3n;      // bits of −(third smallest d² of the un-stored levels); 0 = none; 1 = the query needs the deep pass
    uint32_t slow;
    uint32_t rounds, wave_rounds;  // LOCGPU_STAMP diagnostic build only: main-loop rounds this lane needed / its wave ran
};

typedef uint32_t __attribute__((address_space(3))) lds_u32;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((address_space(3))) lds_u32x2;

__device__ __forceinline__ lds_u32* lds_ptr(uint32_t byte_addr) { return reinterpret_cast<lds_u32*>(byte_addr); }

// First descent (result set empty ⇒ every level pushes). Levels 0..T-1 are not stored: their two smallest d² become rows 0/1, the
// third smallest goes to w.c3n. Levels ≥ T are stored from row 2 on. Leaves w.cur at the first leaf (not yet visited).
// col_addr = LDS byte address of this lane's stack column (row stride ROWB bytes).
// Every lane still in the loop is on the same level (one level per trip from the root), so "is this level stored" is a scalar
// question: two loops — candidates only, then stores only — instead of one body that carries both (the kernel is VALU-bound).
//
// WIN > 0 (round 6): the stored levels are a SLIDING WINDOW — the kernel owns only WIN + 2 rows of LDS (10 rows = 5 120 B per wave =
// 32 waves per CU instead of 21), so of the levels ≥ T only the deepest WIN of a query's own first descent are stored; a level that
// leaves the window joins the un-stored ones (its d² goes through the same candidate insertion, in the same increasing-depth order).
// A query whose leaf lies no deeper than T + WIN — the typical one: T is sized by the tree's DEEPEST branch — stores exactly what the
// fixed scheme stored. The window lives in registers during the descent (the loop is unrolled WIN times, level T + j in pair j mod WIN)
// and is written to rows 2.. in age order behind it.
template <int K, int ROWB, int WIN = 0>
__device__ __forceinline__ void walk_descend(__amdgpu_buffer_rsrc_t rsrc, const uint2* __restrict__ tree, Walk<K>& w, int T, uint32_t col_addr) {
    float c1 = __builtin_inff(), c2 = __builtin_inff(), c3 = __builtin_inff();
    uint32_t f1 = 0, f2 = 0, c1_younger = 0;
    uint32_t cur = 0;
    const float qx = w.qx, qy = w.qy, qz = w.qz;  // values, not lvalues: `c ? w.qx : w.qy` is a select of ADDRESSES and pins w in scratch
    bool at_leaf = false;
    int l = 0;
    // The lanes of a wave are neighbours on a scan ring: they share the top ≈10–14 levels of the tree. While every lane takes the
    // same turn, the node is read ONCE with a scalar load and its axis, threshold and children are wave-uniform — per lane only
    // d² and the candidate update remain (the kernel is VALU-bound, the scalar unit idles).
    {
        uint32_t cur_s = 0;
        for (; l < T; ++l) {
            const uint2 hd = tree[__builtin_amdgcn_readfirstlane(cur_s)];
            const uint32_t meta = __builtin_amdgcn_readfirstlane(hd.y);
            if (meta >= 0xC0000000u) break;  // a leaf above level T: the per-lane loop below sees it
            const float th = as_f32(__builtin_amdgcn_readfirstlane(hd.x));
            const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
            const bool go_left = qa < th;
            const unsigned long long m = __ballot(go_left);
            if (m != 0ull && m != __ballot(true)) break;  // the lanes part here: this level is handled per lane
            const uint32_t right = meta & 0x3FFFFFFFu, cur1 = cur_s + 1u;
            const uint32_t far_slot = m != 0ull ? right : cur1;
            const float dd = qa - th;
            const float d2 = dd * dd;
            const bool lt1 = d2 < c1, lt2 = d2 < c2;
            c3 = __builtin_amdgcn_fmed3f(c2, d2, c3);
            c2 = __builtin_amdgcn_fmed3f(c1, d2, c2);
            c1 = __builtin_fminf(c1, d2);
            const uint32_t f2n = lt2 ? far_slot : f2;
            f2 = lt1 ? f1 : f2n;
            f1 = lt1 ? far_slot : f1;
            const uint32_t yn = lt2 ? 0u : c1_younger;
            c1_younger = lt1 ? 1u : yn;
            cur_s = m != 0ull ? cur1 : right;
        }
        cur = cur_s;
    }
    for (; l < T; ++l) {  // un-stored levels: sorted insertion of d² into (c1 ≤ c2 ≤ c3), far slots of the first two
        const u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
        const uint32_t meta = n.y;
        if (meta >= 0xC0000000u) { at_leaf = true; break; }
        const float th = as_f32(n.x);
        const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
        const float dd = qa - th;
        const float d2 = dd * dd;
        const uint32_t right = meta & 0x3FFFFFFFu;
        const bool go_left = qa < th;
        const uint32_t cur1 = cur + 1u;
        const uint32_t far_slot = go_left ? right : cur1;
        const bool lt1 = d2 < c1, lt2 = d2 < c2;
        c3 = __builtin_amdgcn_fmed3f(c2, d2, c3);
        c2 = __builtin_amdgcn_fmed3f(c1, d2, c2);
        c1 = __builtin_fminf(c1, d2);
        const uint32_t f2n = lt2 ? far_slot : f2;
        f2 = lt1 ? f1 : f2n;
        f1 = lt1 ? far_slot : f1;
        const uint32_t yn = lt2 ? 0u : c1_younger;  // levels come in increasing depth: the new entry is younger than both candidates
        c1_younger = lt1 ? 1u : yn;
        cur = go_left ? cur1 : right;
    }
    int rows = 2;
    if constexpr (WIN > 0) {
        if (!at_leaf) {
            uint32_t wf[WIN], wd[WIN];  // the window: {far slot, bits of −d²} of level T + j in pair j mod WIN
#pragma unroll
            for (int j = 0; j < WIN; ++j) { wf[j] = 0u; wd[j] = 0u; }
            int n3 = 0;        // levels ≥ T this lane has passed
            int pass = 0;      // wave-uniform: the lanes still in the loop are on the same level
            bool found = false;
            for (;;) {
#pragma unroll
                for (int j = 0; j < WIN; ++j) {
                    const u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
                    const uint32_t meta = n.y;
                    if (meta >= 0xC0000000u) { found = true; break; }
                    const float th = as_f32(n.x);
                    const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
                    const float dd = qa - th;
                    const uint32_t right = meta & 0x3FFFFFFFu;
                    const bool go_left = qa < th;
                    const uint32_t cur1 = cur + 1u;
                    if (pass > 0) {  // level T + n3 − WIN leaves the window: an un-stored level like those above T
                        const float d2 = -as_f32(wd[j]);
                        const uint32_t far_slot = wf[j];
                        const bool lt1 = d2 < c1, lt2 = d2 < c2;
                        c3 = __builtin_amdgcn_fmed3f(c2, d2, c3);
                        c2 = __builtin_amdgcn_fmed3f(c1, d2, c2);
                        c1 = __builtin_fminf(c1, d2);
                        const uint32_t f2n = lt2 ? far_slot : f2;
                        f2 = lt1 ? f1 : f2n;
                        f1 = lt1 ? far_slot : f1;
                        const uint32_t yn = lt2 ? 0u : c1_younger;
                        c1_younger = lt1 ? 1u : yn;
                    }
                    wf[j] = go_left ? right : cur1;
                    wd[j] = __float_as_uint(-(dd * dd));
                    n3++;
                    cur = go_left ? cur1 : right;
                }
                if (found) break;
                pass++;
            }
            // rows 2..: the window in age order. Pair j holds the deepest level ≡ j (mod WIN) the lane passed; with more than WIN
            // levels passed the oldest one kept sits in pair n3 mod WIN. Pairs the lane never filled are written too (rows above the
            // stack's top may hold anything).
            const uint32_t rot = n3 > WIN ? (uint32_t)n3 % (uint32_t)WIN : 0u;
            rows = 2 + (n3 < WIN ? n3 : WIN);
            if (__ballot(rot != 0u) == 0ull) {
#pragma unroll
                for (int j = 0; j < WIN; ++j) *reinterpret_cast<lds_u32x2*>(col_addr + (uint32_t)(2 + j) * ROWB) = u32x2{wf[j], wd[j]};
            } else {
#pragma unroll
                for (int j = 0; j < WIN; ++j) {
                    const uint32_t r = (uint32_t)j >= rot ? (uint32_t)j - rot : (uint32_t)j + (uint32_t)WIN - rot;
                    *reinterpret_cast<lds_u32x2*>(col_addr + (2u + r) * ROWB) = u32x2{wf[j], wd[j]};
                }
            }
        }
    } else if (!at_leaf) {
        uint32_t row = col_addr + 2u * ROWB;
        for (;;) {  // stored levels
            const u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
            const uint32_t meta = n.y;
            if (meta >= 0xC0000000u) break;
            const float th = as_f32(n.x);
            const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
            const float dd = qa - th;
            const uint32_t right = meta & 0x3FFFFFFFu;
            const bool go_left = qa < th;
            const uint32_t cur1 = cur + 1u;
            *reinterpret_cast<lds_u32x2*>(row) = u32x2{go_left ? right : cur1, __float_as_uint(-(dd * dd))};
            row += ROWB;
            rows++;
            cur = go_left ? cur1 : right;
        }
    }
    // rows 0 (older) and 1 (younger): popped younger first, each against the bound of its own moment — the recursion's order
    const bool y1 = c1_younger != 0u;
    lds_u32* p0 = lds_ptr(col_addr);
    const uint32_t f_old = y1 ? f2 : f1, f_young = y1 ? f1 : f2;
    const float c_old = y1 ? c2 : c1, c_young = y1 ? c1 : c2;
    p0[0] = f_old;
    p0[1] = __float_as_uint(-c_old);
    p0[ROWB / 4] = f_young;
    p0[ROWB / 4 + 1] = __float_as_uint(-c_young);
    w.cur = cur;
    w.avail = rows;
    w.c3n = c3 < __builtin_inff() ? __float_as_uint(-c3) : 0u;
}

// The main loop in CAPPED ROUNDS. The kernel is bound by VALU issue; a loop in which every lane visits one node per trip pays for the
// leaf block (≈45 instructions) AND the internal block (≈20) on every trip although a lane uses only one of them (round 3: 21.9 ms per
// 256-scan step). Here a round is: a leaf stage for the whole wave (result-set update + pop of up to four rows, to the next node),
// then at most C internal steps per lane; a lane that has not reached its leaf by then sits out the next leaf stage (x = +inf, pop
// switched off) and keeps descending after it (C = 2: 17.3 ms; uncapped rounds, where every round waits for the wave's longest
// descent: 20.0 ms).
//
// A lane is finished when cur == dummy && avail == 0. Latency: a pop only happens on a leaf, and a lane on a leaf pushes nothing — so
// the four youngest stack rows are read at the TOP of the round, next to the node load, instead of behind it.
template <int K, int ROWB, int C, bool STAMP = false>
__device__ __forceinline__ void walk_rounds_capped(__amdgpu_buffer_rsrc_t rsrc, Walk<K>& w, float alpha, uint32_t dummy, uint32_t col_addr, int cap) {
    const float qx = w.qx, qy = w.qy, qz = w.qz;
    uint32_t cur = w.cur;
    int avail = w.avail;
    uint32_t my_rounds = 0, all_rounds = 0;  // STAMP: rounds in which this lane still had work / rounds the wave ran
    float d[K];
    uint32_t id[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { d[j] = w.d[j]; id[j] = w.id[j]; }
    uint32_t slow = w.slow, c3n = w.c3n;
    u32x4 n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
    do {
        if (STAMP) { my_rounds += (cur != dummy || avail > 0) ? 1u : 0u; all_rounds++; }
        const bool is_leaf = n.y >= 0xC0000000u;
        const uint32_t a4 = col_addr + (uint32_t)(avail - 4) * ROWB;
        const u32x2 r3 = *reinterpret_cast<lds_u32x2*>(a4), r2 = *reinterpret_cast<lds_u32x2*>(a4 + ROWB),
                    r1 = *reinterpret_cast<lds_u32x2*>(a4 + 2u * ROWB), r0 = *reinterpret_cast<lds_u32x2*>(a4 + 3u * ROWB);
        const float dx = qx - as_f32(n.x), dy = qy - as_f32(n.z), dz = qz - as_f32(n.w);
        const float dis2 = dx * dx + (dy * dy + dz * dz);
        const float x = is_leaf ? dis2 : __builtin_inff();
        bool c[K];
#pragma unroll
        for (int j = 0; j < K; ++j) c[j] = x < d[j];
        if (K >= 2) {
            const float gap = d[K - 1] - d[K - 2];  // inf − inf = NaN: an empty set is not a tie
            slow = ((gap == 0.0f) & c[K - 1]) ? 1u : slow;  // the AND of two lane masks is scalar work
        }
#pragma unroll
        for (int j = K - 1; j >= 1; --j) {
            const uint32_t below = id[j - 1], here = id[j];
            const uint32_t t = c[j] ? cur : here;
            id[j] = c[j - 1] ? below : t;
            d[j] = __builtin_amdgcn_fmed3f(d[j - 1], x, d[j]);
        }
        {
            const uint32_t here = id[0];
            id[0] = c[0] ? cur : here;
            d[0] = c[0] ? x : d[0];  // = fminf(d[0], x) without its canonicalising v_max (x is never NaN here: `sane`)
        }
        const int nbound = (int)__float_as_uint(-(d[K - 1] * alpha));
        const int nb = is_leaf ? nbound : (int)0x80000000u;
        int hit = 4;
        uint32_t nxt = is_leaf ? dummy : cur;
        hit = (int)r3.y < nb ? 3 : hit; nxt = (int)r3.y < nb ? r3.x : nxt;
        hit = (int)r2.y < nb ? 2 : hit; nxt = (int)r2.y < nb ? r2.x : nxt;
        hit = (int)r1.y < nb ? 1 : hit; nxt = (int)r1.y < nb ? r1.x : nxt;
        hit = (int)r0.y < nb ? 0 : hit; nxt = (int)r0.y < nb ? r0.x : nxt;
        const int avail_eff = is_leaf ? avail : 0;
        const int used = min(min(hit + 1, 4), avail_eff);
        const int low = avail_eff - 1 - min(hit, 3);
        const uint32_t c3sel = low < 2 ? c3n : 0u;
        // Rows 2.. hold one stored level each on the first descent and cannot overflow there; but a CANDIDATE (rows 0/1) sits on an
        // un-stored level, and the descent from it may push more survivors than there are rows (deep, unbalanced trees whose split
        // planes hug the query: a map made of straight lines does it). A push beyond the last row falls off the LDS allocation and
        // is lost, so a stack that stands higher than `cap` rows at a leaf stage — it only grows between two of them — sends the
        // query to the deep pass as well. One compare and a scalar OR per stage.
        const bool deep = ((int)c3sel < nb) | (avail > cap);
        c3n = deep ? 1u : c3n;
        // A query that goes to the deep pass is recomputed there from scratch: its lane RETIRES here (round 4) instead of finishing a
        // traversal whose result is thrown away — these are the long traversals, the ones a wave waits for (0.5-3 % of the queries,
        // but 30-85 % of the waves hold one). After an overflow the rows above the stack's top are not the lane's own either.
        nxt = deep ? dummy : nxt;
        const bool moved = nxt != cur;
        cur = nxt;
        avail = deep ? 0 : avail - used;
        // a lane that popped loads its new node; one that was not on a leaf still holds its internal node in n
        if (moved) n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
#pragma unroll
        for (int step = 0; step < C; ++step) {
            if (n.y < 0xC0000000u) {
                const uint32_t meta = n.y;
                const float th = as_f32(n.x);
                const float qa = meta < 0x40000000u ? qx : (meta < 0x80000000u ? qy : qz);
                const float dd = qa - th;
                const uint32_t nd2 = __float_as_uint(-(dd * dd));
                const uint32_t right = meta & 0x3FFFFFFFu;
                const bool go_left = qa < th;
                const uint32_t cur1 = cur + 1u;
                *reinterpret_cast<lds_u32x2*>(col_addr + (uint32_t)avail * ROWB) = u32x2{go_left ? right : cur1, nd2};
                avail += (int)nd2 < nbound ? 1 : 0;
                cur = go_left ? cur1 : right;
                n = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cur << 3), 0, 0);
            }
        }
    } while (__ballot(cur != dummy || avail > 0) != 0ull);
#pragma unroll
    for (int j = 0; j < K; ++j) { w.d[j] = d[j]; w.id[j] = id[j]; }
    w.slow = slow; w.c3n = c3n; w.cur = cur; w.avail = avail;
    if (STAMP) { w.rounds = my_rounds; w.wave_rounds = all_rounds; }
}

}  // namespace locgpu
