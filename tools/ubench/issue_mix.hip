// tools/ubench/issue_mix.hip — how do VALU and SALU instructions of the SAME SIMD's waves share issue slots on gfx950?
// Each wave runs ITERS iterations of a loop body with V independent VALU adds and S independent SALU adds (inline asm,
// nothing memory-bound). Launched with W waves per SIMD. If a SIMD issued VALU and SALU from different waves in the same
// cycle, time(V,S) ≈ max(time(V,0), time(0,S)); if issue slots are shared, time(V,S) ≈ time(V,0) + time(0,S).
//   hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int V, int S>
__global__ __launch_bounds__(64) void mix_kernel(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < V / 8; ++k)
            asm volatile("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                         "v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#pragma unroll
        for (int k = 0; k < S / 8; ++k)
            asm volatile("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3\n"
                         "s_add_u32 %4, %4, %4\n s_add_u32 %5, %5, %5\n s_add_u32 %6, %6, %6\n s_add_u32 %7, %7, %7"
                         : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f && (s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) == 77u) out[0] = 1.f;
}

template <int V, int S>
static double run(int waves_per_simd, int iters, float* d_out) {
    const int blocks = 256 * 4 * waves_per_simd;  // 256 CUs x 4 SIMDs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix_kernel<V, S>), dim3(blocks), dim3(64), 0, 0, d_out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<V, S>), dim3(blocks), dim3(64), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d_out;
    hipMalloc(&d_out, 4);
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) {
        const double tv = run<64, 0>(w, iters, d_out), ts = run<0, 64>(w, iters, d_out), tm = run<64, 64>(w, iters, d_out), th = run<64, 32>(w, iters, d_out);
        std::printf("waves/SIMD %d: V64 %.3f ms  S64 %.3f ms  V64+S64 %.3f ms  V64+S32 %.3f ms   (sum %.3f, max %.3f)\n", w, tv, ts, tm, th, tv + ts, tv > ts ? tv : ts);
    }
    return 0;
}
