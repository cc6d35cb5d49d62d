// loc_lib_amd/csrc/gn_post.hpp — the finished-scan post of a one-scan alignment paced from the host (locgpu_api.hip, paced_wait;
// icp_kernels.hip, gn_solve_kernel). Plain C++, no HIP in it: the CPU suite drives the host side against injected torn reads
// (tests/cpp/gn_post_sanitize.cpp).
//
// The solve kernel writes, to pinned coherent host memory and WITHOUT a fence (a system-scope release writes the whole L2 back: ≈ 12 µs
// per iteration, measured): first the record of a finished scan, then the 16-byte pair {tag, checksum(tag, record)}. The stores may
// reach the host in any order, so the host takes a record only when the checksum over what IT reads matches the checksum it reads.
//   tag = call << 32 | iterations << 1 | done        (after every iteration; the record and the checksum only with done)
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define LOCGPU_HD __host__ __device__
#else
#define LOCGPU_HD
#endif

namespace locgpu {

struct GnPostRecord {  // what the host needs of a finished scan's PoseState
    static constexpr int kWords = 10;
    unsigned long long w[kWords];  // q[4], t[3], last_dx_norm (bits) | last_eff | converged << 32 | status
};
struct GnPost {
    GnPostRecord* record = nullptr;      // pinned, coherent host memory, 16-byte aligned
    unsigned long long* word = nullptr;  // 16-byte aligned: [0] call << 32 | iterations << 1 | done, after every iteration; [1] checksum sealing the record
    unsigned int call = 0;
};
// the checksum of a posted record (kernel and host compute the same)
LOCGPU_HD inline unsigned long long gn_post_sum(unsigned long long tag, const GnPostRecord& r) {
    unsigned long long sum = tag * 0x9e3779b97f4a7c15ull;
    for (int i = 0; i < GnPostRecord::kWords; ++i) sum = (sum ^ r.w[i]) * 0x100000001b3ull;
    return sum;
}

// Host side: is there a post of call `call` that is newer than iteration `seen`? `area` = the record's words followed (at word_at) by
// {tag, checksum}. Returns false while there is none — an older call's word, an iteration already seen, or a done post whose record
// is still on its way (torn: the checksum over the words read does not match). On true *tag_out is the tag and, when its done bit is
// set, *rec the record that belongs to it.
inline bool gn_post_take(const unsigned long long* area, int word_at, unsigned int call, int seen, unsigned long long* tag_out, GnPostRecord* rec) {
    const unsigned long long* word = area + word_at;
    const unsigned long long w = __atomic_load_n(word, __ATOMIC_ACQUIRE);
    if ((unsigned int)(w >> 32) != call || (int)((w & 0xffffffffull) >> 1) <= seen) return false;
    if (w & 1ull) {
        for (int i = 0; i < GnPostRecord::kWords; ++i) rec->w[i] = __atomic_load_n(area + i, __ATOMIC_RELAXED);
        if (gn_post_sum(w, *rec) != __atomic_load_n(word + 1, __ATOMIC_RELAXED)) return false;  // still on its way
    }
    *tag_out = w;
    return true;
}

}  // namespace locgpu
