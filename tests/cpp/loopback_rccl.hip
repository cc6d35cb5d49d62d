// tests/cpp/loopback_rccl.hip — TEST DOUBLE of the six RCCL entry points liblocgpu.so binds (csrc/locgpu_api.hip: rccl()), for ranks
// that are THREADS of one process sharing one GPU. RCCL itself refuses two ranks on one device ("Duplicate GPU detected") and
// the boxes this suite runs on have one GPU, so this is the only way to drive the library's world-size-2 control flow — scan
// shards, the owner solving ahead of the exchange, the ring of exchange buffers, collectives of two batches in flight on the
// communication stream, the tree broadcast and its status exchange — on real hardware. Selected with LOCGPU_RCCL_LIB=<this .so>.
//
// Semantics kept: a collective is ordered on the stream it is given, on every rank; results are the elementwise reduction in rank
// order. Semantics NOT kept: the call itself blocks the calling host thread until every rank of the communicator has made the
// matching call (RCCL only enqueues). That is a stricter schedule than the real one, never a looser one: the ranks make the same
// sequence of collectives, so a host rendezvous per collective cannot deadlock unless the library's own order differs between
// ranks — which is exactly the bug this double exists to expose (the rendezvous times out after 60 s and returns an error).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {
constexpr int kRing = 4;  // staging slots per rank: a slot is reused four collectives later, ordered by the peers' `done` events

struct Group;
struct Comm {
    Group* g = nullptr;
    int rank = 0;
    uint64_t seq = 0;
    void* stage[kRing] = {};
    size_t stage_bytes[kRing] = {};
    hipEvent_t ready[kRing] = {};  // this rank's contribution of the slot is in `stage`
    hipEvent_t done[kRing] = {};   // this rank has finished reading everybody's `stage` of the slot
};
struct Group {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    std::vector<Comm*> comms;
    int joined = 0;
    // rendezvous state of the collective in progress
    uint64_t seq = 0;
    int arrived = 0;
    int left = 0;
    size_t bytes = 0;
    bool mismatch = false;
};
std::mutex g_mu;
std::map<uint64_t, std::unique_ptr<Group>> g_groups;
std::atomic<unsigned long long> g_calls[8];  // collectives entered per rank, over all communicators of the process (loopback_rccl_counts)
std::atomic<uint64_t> g_next_id{1};

constexpr int kMaxWorld = 8;
struct Parts { const void* p[kMaxWorld]; };

template <typename T, int OP>  // OP 0 = sum, 1 = min
__global__ void reduce_kernel(T* out, Parts parts, int world, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T acc = ((const T*)parts.p[0])[i];
    for (int r = 1; r < world; ++r) {
        const T v = ((const T*)parts.p[r])[i];
        acc = OP == 0 ? acc + v : (v < acc ? v : acc);
    }
    out[i] = acc;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclChar: case ncclUint8: return 1;
        case ncclInt: case ncclUint32: case ncclFloat: return 4;
        case ncclInt64: case ncclUint64: case ncclDouble: return 8;
        default: return 0;
    }
}

// Every rank: wait until all ranks have entered collective number `seq` of the group with the same byte count, having run
// `before` (which records this rank's `ready` event) first; then run `after` (which may wait on the peers' events).
template <typename Before, typename After>
ncclResult_t rendezvous(Comm* c, size_t bytes, Before before, After after) {
    Group* g = c->g;
    const uint64_t seq = c->seq++;
    g_calls[c->rank & 7]++;
    const int slot = (int)(seq % kRing);
    if (!before(slot)) return ncclUnhandledCudaError;
    {
        std::unique_lock<std::mutex> lk(g->m);
        // the previous collective must have been left by everybody before this one starts counting
        if (!g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->seq == seq && g->left == 0; })) {
            std::fprintf(stderr, "loopback_rccl: rank %d stuck before collective %llu (group at %llu)\n", c->rank, (unsigned long long)seq, (unsigned long long)g->seq);
            return ncclInternalError;
        }
        if (g->arrived == 0) { g->bytes = bytes; g->mismatch = false; }
        else if (g->bytes != bytes) g->mismatch = true;
        g->arrived++;
        g->cv.notify_all();
        if (!g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->arrived == g->world; })) {
            std::fprintf(stderr, "loopback_rccl: rank %d alone in collective %llu (%d of %d arrived)\n", c->rank, (unsigned long long)seq, g->arrived, g->world);
            return ncclInternalError;
        }
        if (g->mismatch) {
            std::fprintf(stderr, "loopback_rccl: collective %llu called with different sizes on different ranks\n", (unsigned long long)seq);
            return ncclInvalidArgument;
        }
    }
    const bool ok = after(slot);
    {
        std::unique_lock<std::mutex> lk(g->m);
        if (++g->left == g->world) { g->left = 0; g->arrived = 0; g->seq = seq + 1; }
        g->cv.notify_all();
    }
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

bool ensure_stage(Comm* c, int slot, size_t bytes, hipStream_t s) {
    // The slot's previous use (kRing collectives ago) must have been read by every peer before it is overwritten or freed.
    for (Comm* p : c->g->comms)
        if (p != c && hipStreamWaitEvent(s, p->done[slot], 0) != hipSuccess) return false;
    if (c->stage_bytes[slot] < bytes) {
        if (c->stage[slot]) {
            for (Comm* p : c->g->comms)
                if (p != c && hipEventSynchronize(p->done[slot]) != hipSuccess) return false;
            if (hipFree(c->stage[slot]) != hipSuccess) return false;
        }
        if (hipMalloc(&c->stage[slot], bytes) != hipSuccess) return false;
        c->stage_bytes[slot] = bytes;
    }
    return true;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof(*id));
    const uint64_t v = g_next_id++;
    std::memcpy(id->internal, &v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    uint64_t key;
    std::memcpy(&key, id.internal, sizeof(key));
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto& slot = g_groups[key];
        if (!slot) { slot.reset(new Group); slot->world = world; slot->comms.assign(world, nullptr); }
        g = slot.get();
    }
    if (g->world != world || rank < 0 || rank >= world || world > kMaxWorld) return ncclInvalidArgument;
    Comm* c = new Comm;
    c->g = g;
    c->rank = rank;
    for (int i = 0; i < kRing; ++i) {
        if (hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    std::unique_lock<std::mutex> lk(g->m);
    if (g->comms[rank]) return ncclInvalidArgument;
    g->comms[rank] = c;
    g->joined++;
    g->cv.notify_all();
    if (!g->cv.wait_for(lk, std::chrono::seconds(60), [&] { return g->joined == g->world; })) return ncclInternalError;
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < kRing; ++i) {
        if (c->stage[i]) (void)hipFree(c->stage[i]);
        (void)hipEventDestroy(c->ready[i]);
        (void)hipEventDestroy(c->done[i]);
    }
    // the Group keeps the pointer slot (a destroyed rank makes no further collectives); the Comm itself is leaked on purpose so a
    // peer still inside its last rendezvous never touches freed memory
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = count * type_bytes(type);
    const bool is_double_sum = type == ncclDouble && op == ncclSum, is_int_min = type == ncclInt && op == ncclMin;
    if (!c || !bytes || !(is_double_sum || is_int_min)) return ncclInvalidArgument;
    if (c->g->world == 1) {
        if (send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
        c->seq++;
        return ncclSuccess;
    }
    return rendezvous(
        c, bytes,
        [&](int slot) {
            return ensure_stage(c, slot, bytes, s) &&
                   hipMemcpyAsync(c->stage[slot], send, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess && hipEventRecord(c->ready[slot], s) == hipSuccess;
        },
        [&](int slot) {
            Group* g = c->g;
            Parts parts{};
            for (int r = 0; r < g->world; ++r) {
                parts.p[r] = g->comms[r]->stage[slot];
                if (r != c->rank && hipStreamWaitEvent(s, g->comms[r]->ready[slot], 0) != hipSuccess) return false;
            }
            const unsigned blocks = (unsigned)((count + 255) / 256);
            if (is_double_sum) reduce_kernel<double, 0><<<blocks, 256, 0, s>>>((double*)recv, parts, g->world, count);
            else reduce_kernel<int, 1><<<blocks, 256, 0, s>>>((int*)recv, parts, g->world, count);
            return hipGetLastError() == hipSuccess && hipEventRecord(c->done[slot], s) == hipSuccess;
        });
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t comm, hipStream_t s) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = count * type_bytes(type);
    if (!c || root < 0 || root >= c->g->world) return ncclInvalidArgument;
    if (c->g->world == 1 || bytes == 0) {
        if (bytes && send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return ncclUnhandledCudaError;
        if (c->g->world == 1) { c->seq++; return ncclSuccess; }
    }
    return rendezvous(
        c, bytes,
        [&](int slot) {
            if (c->rank != root) {
                // non-roots contribute nothing, but their slot's previous readers must still be ordered before `done` is re-recorded
                for (Comm* p : c->g->comms)
                    if (p != c && hipStreamWaitEvent(s, p->done[slot], 0) != hipSuccess) return false;
                return hipEventRecord(c->ready[slot], s) == hipSuccess;
            }
            return ensure_stage(c, slot, bytes ? bytes : 8, s) && (bytes == 0 || hipMemcpyAsync(c->stage[slot], send, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess) &&
                   hipEventRecord(c->ready[slot], s) == hipSuccess;
        },
        [&](int slot) {
            Comm* r = c->g->comms[root];
            if (c->rank != root) {
                if (hipStreamWaitEvent(s, r->ready[slot], 0) != hipSuccess) return false;
                if (bytes && hipMemcpyAsync(recv, r->stage[slot], bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) return false;
            } else if (bytes && send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                return false;
            }
            return hipEventRecord(c->done[slot], s) == hipSuccess;
        });
}

// test hook: collectives each rank has entered so far (the ranks of a correct library make the same number of them)
void loopback_rccl_counts(unsigned long long out[8]) {
    for (int r = 0; r < 8; ++r) out[r] = g_calls[r].load();
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "loopback: HIP call failed";
        case ncclInternalError: return "loopback: a rank did not reach the collective (order differs between ranks?)";
        case ncclInvalidArgument: return "loopback: invalid argument or mismatched sizes";
        default: return "loopback: error";
    }
}
}  // extern "C"
