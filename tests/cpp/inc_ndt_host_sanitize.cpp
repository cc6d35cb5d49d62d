// tests/cpp/inc_ndt_host_sanitize.cpp — the host side of the incremental-NDT target ingest (loc_lib_amd/csrc/inc_ndt_lru.hpp), meant to be
// compiled with -fsanitize=address,undefined (tests/test_abi_and_host.py; GPU sanitizers do not exist on this pool). Two things are checked on
// random multi-call key sequences with capacities small enough to evict:
//   (1) inc_lru_replay — the per-point replay the library uses when a cloud's own working set exceeds the capacity — against a plain
//       restatement of NdtRegistration::SetIncNdtTargetCloud's loop (ndt_registration.cpp:150-171: std::list of {key, points} + a map to list
//       nodes): same voxel set, same recency order, same surviving points per voxel;
//   (2) the closed form the DEVICE path uses when the cloud touches m <= capacity - 1 distinct voxels (ndt_inc.hip): nothing the call touches is
//       evicted, every touched voxel keeps all its points of the call, and the voxels that leave are exactly the E = max(0, live + new - (capacity - 1))
//       old voxels with the smallest recency stamps — independent of the order of the points.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <list>
#include <map>
#include <random>
#include <set>
#include <utility>
#include <vector>

#include "inc_ndt_lru.hpp"

using locgpu::IncLive;

namespace {

// The reference's container, restated naively: front = most recent. Each voxel holds the indices of the points added to this INSTANCE
// during the current call (AddPoint / the constructor's {pt}); UpdateVoxel clears them at the end of every call.
struct RefVoxel { uint64_t key; std::vector<size_t> pts; };
struct RefModel {
    std::list<RefVoxel> data;
    std::map<uint64_t, std::list<RefVoxel>::iterator> grids;
    size_t capacity;
    void call(const std::vector<uint64_t>& keys) {
        for (auto& v : data) v.pts.clear();
        for (size_t i = 0; i < keys.size(); ++i) {
            if (keys[i] == locgpu::kIncNoKey) continue;
            auto it = grids.find(keys[i]);
            if (it == grids.end()) {
                data.push_front(RefVoxel{keys[i], {i}});
                grids[keys[i]] = data.begin();
                if (data.size() >= capacity) { grids.erase(data.back().key); data.pop_back(); }
            } else {
                it->second->pts.push_back(i);
                data.splice(data.begin(), data, it->second);
                it->second = data.begin();
            }
        }
    }
};

int fail(const char* what, int trial, int call) { std::printf("FAILED (%s) in trial %d, call %d\n", what, trial, call); return 1; }

}  // namespace

int main() {
    std::mt19937_64 rng(20241004);
    int replayed_calls = 0, closed_form_calls = 0;
    for (int trial = 0; trial < 300; ++trial) {
        const size_t capacity = 2 + rng() % (trial % 3 == 0 ? 6 : 60);
        const uint64_t universe = 1 + rng() % (3 * capacity + 5);
        RefModel ref;
        ref.capacity = capacity;
        std::vector<IncLive> live;
        std::vector<int> free_slots;
        int n_slots = 0;
        for (int call = 1; call <= 6; ++call) {
            const size_t n = rng() % 400;
            std::vector<uint64_t> keys(n);
            const bool runs = rng() % 2;  // clouds have locality: runs of points in one voxel
            for (size_t i = 0; i < n; ++i) {
                if (runs && i > 0 && rng() % 4 != 0) keys[i] = keys[i - 1];
                else keys[i] = rng() % 23 == 0 ? locgpu::kIncNoKey : 1000 + rng() % universe;
            }
            // ---- (2) closed form, from the state BEFORE the call
            std::set<uint64_t> touched;
            std::map<uint64_t, uint64_t> last_idx;
            for (size_t i = 0; i < n; ++i) if (keys[i] != locgpu::kIncNoKey) { touched.insert(keys[i]); last_idx[keys[i]] = i; }
            const size_t M = capacity - 1, m = touched.size();
            std::vector<std::pair<uint64_t, uint64_t>> predicted;  // {stamp, key} of the voxel set the device path would produce
            if (m <= M) {
                std::vector<std::pair<uint64_t, uint64_t>> old_untouched;
                size_t m_new = m;
                for (const IncLive& v : live) {
                    if (touched.count(v.key)) m_new--;
                    else old_untouched.push_back({v.stamp, v.key});
                }
                std::sort(old_untouched.begin(), old_untouched.end());
                const size_t total = live.size() + m_new;
                const size_t n_evict = total > M ? total - M : 0;
                if (n_evict > old_untouched.size()) return fail("closed form wants to evict a touched voxel", trial, call);
                for (size_t j = n_evict; j < old_untouched.size(); ++j) predicted.push_back(old_untouched[j]);
                for (uint64_t k : touched) predicted.push_back({((uint64_t)call << 32) | last_idx[k], k});
                std::sort(predicted.begin(), predicted.end());
            }
            // ---- (1) the replay and the reference model
            std::vector<unsigned char> keep;
            locgpu::inc_lru_replay(live, free_slots, n_slots, capacity, (uint32_t)call, keys.data(), n, keep);
            ref.call(keys);
            replayed_calls++;
            if (live.size() != ref.data.size()) return fail("voxel count", trial, call);
            if (live.size() >= capacity) return fail("more voxels than the list can hold", trial, call);
            auto rit = ref.data.begin();
            std::vector<unsigned char> ref_keep(n, 0);
            std::set<int> slots_in_use;
            for (size_t j = 0; j < live.size(); ++j, ++rit) {  // both most recent first
                if (live[j].key != rit->key) return fail("recency order", trial, call);
                if (j > 0 && !(live[j - 1].stamp > live[j].stamp)) return fail("stamps do not order the list", trial, call);
                if (live[j].slot < 0 || live[j].slot >= n_slots || !slots_in_use.insert(live[j].slot).second) return fail("slot handed out twice", trial, call);
                for (size_t i : rit->pts) ref_keep[i] = 1;
                if (!rit->pts.empty() && live[j].stamp != (((uint64_t)call << 32) | rit->pts.back())) return fail("stamp of a touched voxel", trial, call);
            }
            for (int sl : free_slots) if (sl < 0 || sl >= n_slots || !slots_in_use.insert(sl).second) return fail("free slot also in use", trial, call);
            if ((int)slots_in_use.size() != n_slots) return fail("a slot is neither live nor free", trial, call);
            if (keep != ref_keep) return fail("surviving points", trial, call);
            if (m <= M) {
                closed_form_calls++;
                std::vector<std::pair<uint64_t, uint64_t>> got;
                for (const IncLive& v : live) got.push_back({v.stamp, v.key});
                std::sort(got.begin(), got.end());
                if (got != predicted) return fail("closed form != sequential replay", trial, call);
                for (size_t i = 0; i < n; ++i)
                    if (keep[i] != (keys[i] != locgpu::kIncNoKey)) return fail("closed form: a touched voxel lost points", trial, call);
            }
        }
    }
    std::printf("inc-ndt host harness ok: %d replayed calls equal the reference loop, %d of them also equal the closed form\n", replayed_calls, closed_form_calls);
    return 0;
}
