// tests/cpp/lds_semantics.hip — pins the hardware behaviour the hot search kernel relies on (loc_lib_amd/csrc/search_walk.hpp):
//   (1) with no static __shared__ in the kernel, the dynamic LDS block of a one-wave workgroup starts at LDS address 0;
//   (2) a ds_read of an address BELOW 0 (wrapped: 0xFFFFF...) returns 0 — also when many other workgroups, whose LDS is full of
//       non-zero words, are resident on the same CU;
//   (3) a ds_write at or above the end of the allocation changes neither this workgroup's words nor a neighbour's: it is dropped,
//       or it lands in the padding up to the workgroup's allocation granule (1280 B on gfx950: a 6144-byte request owns 6400).
// The walk kernels read stack rows avail-4..avail-1 unconditionally (rows below the bottom must read {0,0}) and push
// unconditionally to the row above the top (a push beyond the last row must be harmless; the kernel notices the overflow by
// itself). They never READ above the top, so what such a read returns is reported but not required (it is 0 beyond the
// granule and whatever was stored there inside it). Built by __graft_entry__.build(), run by tests/test_gpu_lds_semantics.py
// on the GPU box. Exit status 0 = every assumption holds; each violated one is printed.
//
//     hipcc --offload-arch=gfx950 -O2 lds_semantics.hip -o lds_semantics && ./lds_semantics
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t __attribute__((address_space(3))) lds_u32;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 __attribute__((address_space(3))) lds_u32x2;

// One-wave workgroups like the search kernel's; `bytes` of dynamic LDS each. Every workgroup fills its block with a pattern that
// names the workgroup, waits long enough for its CU to be full of such blocks, then probes outside and checks nobody wrote into it.
// out[wg] = bit mask of violated assumptions.
__global__ __launch_bounds__(64) void probe(uint32_t* __restrict__ out, int bytes, int spin) {
    extern __shared__ uint32_t s[];
    const int tid = threadIdx.x;
    const uint32_t pat = 0xA5000000u | ((uint32_t)blockIdx.x << 8);
    for (int i = tid; i < bytes / 4; i += 64) s[i] = pat + (uint32_t)(i & 255);
    __syncthreads();
    const uint32_t base = (uint32_t)(size_t)s;
    uint32_t bad = base != 0u ? 1u : 0u;                                     // (1)
    // let the dispatcher fill the CU: neighbours' blocks are live (and non-zero) while this one probes
    for (volatile int k = 0; k < spin; ++k) {}
    // (2) reads: one 8-byte row per lane, like the kernel's lds_u32x2 accesses, below the bottom and above the top
    const uint32_t col = base + (uint32_t)tid * 8u;
    for (int r = 1; r <= 4; ++r) {
        const u32x2 v = *reinterpret_cast<lds_u32x2*>(col - (uint32_t)r * 512u);  // rows -1..-4 of a 64-lane stack
        bad |= (v.x | v.y) != 0u ? 2u : 0u;
    }
    for (int r = 0; r < 4; ++r) {
        const u32x2 v = *reinterpret_cast<lds_u32x2*>(col + (uint32_t)bytes + (uint32_t)r * 512u);  // the rows above the last
        bad |= (v.x | v.y) != 0u ? 4u : 0u;
    }
    {
        const u32x2 v = *reinterpret_cast<lds_u32x2*>(col + 65536u);
        const uint32_t w = *reinterpret_cast<lds_u32*>(base + 163840u - 4u);
        bad |= ((v.x | v.y) != 0u || (w != 0u && bytes < 163840)) ? 8u : 0u;
    }
    // (3) writes outside: dropped
    *reinterpret_cast<lds_u32x2*>(col + (uint32_t)bytes) = u32x2{0xDEAD0001u, 0xDEAD0002u};
    *reinterpret_cast<lds_u32x2*>(col + (uint32_t)bytes + 2048u) = u32x2{0xDEAD0003u, 0xDEAD0004u};
    *reinterpret_cast<lds_u32x2*>(col - 512u) = u32x2{0xDEAD0005u, 0xDEAD0006u};
    __syncthreads();
    for (volatile int k = 0; k < spin; ++k) {}
    {
        const u32x2 v = *reinterpret_cast<lds_u32x2*>(col + (uint32_t)bytes);
        const u32x2 u = *reinterpret_cast<lds_u32x2*>(col - 512u);
        bad |= ((v.x | v.y | u.x | u.y) != 0u) ? 16u : 0u;                  // still reads 0 after the write
    }
    uint32_t corrupt = 0;
    for (int i = tid; i < bytes / 4; i += 64) corrupt |= s[i] != pat + (uint32_t)(i & 255) ? 1u : 0u;  // nobody's stray write landed here
    bad |= corrupt ? 32u : 0u;
    // wave OR
    for (int off = 32; off > 0; off >>= 1) bad |= __shfl_xor(bad, off, 64);
    if (tid == 0) out[blockIdx.x] = bad;
}

int main() {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { printf("LDS SEMANTICS: no HIP device\n"); return 2; }
    const int n_wg = 8192;  // 256 CUs x up to 21 resident one-wave workgroups, several rounds
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, n_wg * sizeof(uint32_t)) != hipSuccess) return 2;
    static const char* what[6] = {"dynamic LDS does not start at address 0", "a read below the allocation returned non-zero",
                                  "a read just above the requested size returned non-zero", "a read far above the allocation returned non-zero",
                                  "a write just outside the requested size reads back",
                                  "a workgroup's own LDS words were changed by somebody's out-of-range write"};
    const uint32_t required = 1u | 2u | 8u | 32u;  // bits 4 and 16 are informative: inside the allocation granule the padding is ordinary LDS
    int rc = 0;
    // the search kernels' sizes: 15 / 12 / 24 rows x 64 lanes x 8 B, the 34-row deep pass, the 16-lane one-scan kernel (34 x 128 B)
    for (int bytes : {15 * 512, 12 * 512, 24 * 512, 34 * 512, 34 * 128, 1280}) {
        (void)hipMemset(d, 0xFF, n_wg * sizeof(uint32_t));
        hipLaunchKernelGGL(probe, dim3(n_wg), dim3(64), bytes, 0, d, bytes, 2000);
        if (hipDeviceSynchronize() != hipSuccess) { printf("LDS SEMANTICS: launch failed at %d bytes: %s\n", bytes, hipGetErrorString(hipGetLastError())); return 2; }
        std::vector<uint32_t> h(n_wg);
        (void)hipMemcpy(h.data(), d, n_wg * sizeof(uint32_t), hipMemcpyDeviceToHost);
        uint32_t any = 0;
        int n_bad = 0;
        for (uint32_t v : h) { any |= v; n_bad += v != 0u; }
        printf("LDS SEMANTICS: %6d B per workgroup, %d workgroups: %s (mask %#x, %d workgroups)\n", bytes, n_wg, (any & required) ? "VIOLATED" : "ok", any, n_bad);
        for (int b = 0; b < 6; ++b)
            if (any & (1u << b)) {
                printf("  - %s%s\n", what[b], (required & (1u << b)) ? "" : " (not required: padding of the 1280-byte allocation granule)");
                if (required & (1u << b)) rc = 1;
            }
    }
    (void)hipFree(d);
    printf(rc ? "LDS SEMANTICS FAILED\n" : "LDS SEMANTICS OK\n");
    return rc;
}
