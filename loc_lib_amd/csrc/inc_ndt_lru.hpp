// loc_lib_amd/csrc/inc_ndt_lru.hpp — the sequential form of NdtRegistration::SetIncNdtTargetCloud's voxel bookkeeping
// (ndt_registration.cpp:150-171), plain C++ with no HIP in it so that it also builds under the host sanitizers
// (tests/cpp/inc_ndt_host_sanitize.cpp).
//
// The reference keeps its voxels in a std::list ordered by recency (front = most recently touched) plus a hash map key → list
// node. Per point, in input order: a known voxel is moved to the front (:169-170); an unknown one is pushed to the front and,
// once the list has reached `capacity_` entries, the voxel at the back is erased (:158-165). That is an LRU cache that holds at
// most capacity − 1 voxels after every point.
//
// liblocgpu keeps the voxel set on the device as per-slot arrays (key, recency stamp, μ, info) and replays this loop on the device
// WITHOUT walking the points in order whenever the cloud touches at most capacity − 1 distinct voxels (ndt_inc.hip: an LRU cache
// always holds the most recently used keys, and then no voxel touched by the call can be evicted again within it). Only a cloud
// whose own working set exceeds the capacity — test-sized capacities — needs the order of the points: that case is replayed here,
// on the host, from the device's state.
//
// Recency stamps: (call number << 32) | index of the last point of that call that touched the voxel. Larger = more recent; the
// reference's list order is the descending stamp order.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <algorithm>
#include <list>
#include <unordered_map>
#include <vector>

namespace locgpu {

constexpr uint64_t kIncNoKey = ~0ull;  // a point outside the key range (= kNdtEmpty): it touches no voxel

struct IncLive {  // one live voxel of the device state
    uint64_t key, stamp;
    int slot;
};

struct IncKeyHash {
    size_t operator()(uint64_t k) const {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
        return (size_t)k;
    }
};

// Replays `keys[0..n)` (one per point, kIncNoKey = skip) on the state {live, free_slots, n_slots}. keep[i] = 1 iff point i belongs to
// a voxel instance that is still alive at the end of the call (points of an instance evicted later within the call are lost with it,
// ndt cpp:161-165; the reference then dereferences a stale entry for such voxels, :178 — they are simply not updated here).
// Slots freed during the call are not handed out again before its end.
inline void inc_lru_replay(std::vector<IncLive>& live, std::vector<int>& free_slots, int& n_slots, size_t capacity, uint32_t epoch, const uint64_t* keys,
                           size_t n, std::vector<unsigned char>& keep) {
    struct Node { uint64_t key, stamp; int slot; };
    std::list<Node> lru;  // front = most recent
    std::sort(live.begin(), live.end(), [](const IncLive& a, const IncLive& b) { return a.stamp > b.stamp; });
    for (const IncLive& v : live) lru.push_back(Node{v.key, v.stamp, v.slot});
    std::unordered_map<uint64_t, std::list<Node>::iterator, IncKeyHash> map;
    map.reserve(2 * (lru.size() + 16));
    for (auto it = lru.begin(); it != lru.end(); ++it) map.emplace(it->key, it);
    std::vector<int> pt_slot(n, -1);
    std::vector<int> died;
    for (size_t i = 0; i < n; ++i) {
        const uint64_t key = keys[i];
        if (key == kIncNoKey) continue;
        const uint64_t stamp = ((uint64_t)epoch << 32) | (uint64_t)(uint32_t)i;
        auto it = map.find(key);
        if (it == map.end()) {
            int slot;
            if (!free_slots.empty()) { slot = free_slots.back(); free_slots.pop_back(); }
            else slot = n_slots++;
            lru.push_front(Node{key, stamp, slot});
            map.emplace(key, lru.begin());
            pt_slot[i] = slot;
            if (lru.size() >= capacity) {  // ndt cpp:161-165: drop the least recently used voxel
                died.push_back(lru.back().slot);
                map.erase(lru.back().key);
                lru.pop_back();
            }
        } else {
            it->second->stamp = stamp;
            lru.splice(lru.begin(), lru, it->second);  // touched ⇒ most recent (ndt cpp:169-170); the iterator stays valid
            pt_slot[i] = it->second->slot;
        }
    }
    std::vector<unsigned char> dead((size_t)std::max(n_slots, 1), 0);
    for (int sl : died) dead[(size_t)sl] = 1;
    keep.assign(n, 0);
    for (size_t i = 0; i < n; ++i) keep[i] = pt_slot[i] >= 0 && !dead[(size_t)pt_slot[i]];
    for (int sl : died) free_slots.push_back(sl);
    live.clear();
    for (const Node& v : lru) live.push_back(IncLive{v.key, v.stamp, v.slot});
}

}  // namespace locgpu
