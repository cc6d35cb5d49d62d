// loc_lib_amd/csrc/pool_sched.hpp — the bookkeeping of the open-scan pool (scan_pool.hip): which job holds which source regions, which
// scan sits in which slot, who waits. Plain C++ with no HIP in it, so that the CPU suite runs it under AddressSanitizer / UBSan and
// checks the property the several-GPU mode rests on (tests/cpp/pool_sched_sanitize.cpp): EVERY DECISION IS A FUNCTION OF THE ORDER OF THE
// CALLS AND OF THE FLAGS ALL RANKS SEE — never of which scans a rank happens to hold, and never of a copy's timing. Two ranks that make
// the same calls and read the same convergence flags therefore give every scan the same slot and the same region at the same chunk
// boundary, and issue the same number of collectives.
#pragma once
#include <algorithm>
#include <cstdint>
#include <deque>
#include <functional>
#include <vector>

namespace locgpu {

struct PoolSchedJob {
    int64_t ticket = 0;
    int n_total = 0;          // scans of the job (all ranks')
    int first = 0, n_local = 0;  // this rank holds the points of scans [first, first + n_local)
    std::vector<int> region;  // [n_total] source region of every scan (a rank only fills the ones it holds)
    int next = 0;             // scans [0, next) have been given a slot
    int remaining = 0;        // scans not finished yet
    bool holds(int i) const { return i >= first && i < first + n_local; }
};

struct PoolAdmitted {
    int slot;
    PoolSchedJob* job;
    int idx;  // scan of the job
};

class PoolSched {
public:
    PoolSched(int slots, int regions) : slot_job_((size_t)slots, nullptr), slot_idx_((size_t)slots, 0) {
        free_slots_.resize((size_t)slots);
        for (int s = 0; s < slots; ++s) free_slots_[(size_t)s] = slots - 1 - s;  // descending: pop_back() hands out the smallest
        free_regions_.resize((size_t)regions);
        for (int r = 0; r < regions; ++r) free_regions_[(size_t)r] = regions - 1 - r;
    }
    int slots() const { return (int)slot_job_.size(); }
    int free_slots() const { return (int)free_slots_.size(); }
    int free_regions() const { return (int)free_regions_.size(); }
    size_t waiting() const { return waiting_.size(); }
    PoolSchedJob* job_of(int slot) const { return slot_job_[(size_t)slot]; }
    int idx_of(int slot) const { return slot_idx_[(size_t)slot]; }

    // A job needs n_total regions — on EVERY rank, whichever of its scans the rank holds. False: not enough free ones.
    bool accept(PoolSchedJob* j) {
        if (j->n_total < 1 || j->n_total > free_regions()) return false;
        j->region.resize((size_t)j->n_total);
        for (int i = 0; i < j->n_total; ++i) { j->region[(size_t)i] = free_regions_.back(); free_regions_.pop_back(); }
        j->next = 0;
        j->remaining = j->n_total;
        waiting_.push_back(j);
        return true;
    }
    // The job accepted LAST is taken back (its copy could not be started): its regions return.
    void cancel_last(PoolSchedJob* j) {
        if (waiting_.empty() || waiting_.back() != j || j->next != 0) return;
        waiting_.pop_back();
        for (int i = 0; i < j->n_total; ++i) put_back(free_regions_, j->region[(size_t)i]);
    }
    // Waiting scans enter one by one, oldest job first, while there are free slots. ready(job, idx) is asked before a scan is given a
    // slot: false = not now (its points are still on their way) — admission stops there, later jobs do not overtake. With several
    // ranks `ready` must not depend on anything rank-local: it always says true and the stream waits for the copy instead.
    template <class Ready>
    void admit(Ready&& ready, std::vector<PoolAdmitted>& out) {
        while (!waiting_.empty() && !free_slots_.empty()) {
            PoolSchedJob* j = waiting_.front();
            bool stalled = false;
            while (j->next < j->n_total && !free_slots_.empty()) {
                if (!ready(j, j->next)) { stalled = true; break; }
                const int i = j->next++;
                const int sl = free_slots_.back();
                free_slots_.pop_back();
                slot_job_[(size_t)sl] = j;
                slot_idx_[(size_t)sl] = i;
                out.push_back(PoolAdmitted{sl, j, i});
            }
            if (j->next == j->n_total) waiting_.pop_front();
            if (stalled) break;
        }
    }
    // The scan in `slot` has finished: slot and region are free again.
    void finish(int slot) {
        PoolSchedJob* j = slot_job_[(size_t)slot];
        if (!j) return;
        j->remaining--;
        put_back(free_regions_, j->region[(size_t)slot_idx_[(size_t)slot]]);
        slot_job_[(size_t)slot] = nullptr;
        put_back(free_slots_, slot);
    }

private:
    static void put_back(std::vector<int>& v, int x) { v.insert(std::upper_bound(v.begin(), v.end(), x, std::greater<int>()), x); }
    std::vector<int> free_slots_, free_regions_;  // kept sorted descending
    std::vector<PoolSchedJob*> slot_job_;         // nullptr = free
    std::vector<int> slot_idx_;
    std::deque<PoolSchedJob*> waiting_;           // accepted, not (completely) admitted yet: FIFO
};

}  // namespace locgpu
