// FP64 vector issue rate of the MI355X as the fit kernel sees it: independent FMA chains (ILP 1..8) at 1..8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 fp64_rate.hip -o fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b) {
    double x[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    if (s == 12345.678) out[0] = s;
}
template <int ILP>
void run(double* d, int waves_per_simd) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;  // 256 CUs x 4 SIMDs x waves / 4 waves per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0000001, 1e-9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<ILP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fmas = (double)blocks * 256 * iters * ILP;
    printf("ILP %d, %d waves/SIMD: %.2f TFLOP/s FP64, %.2f cycles per wave64 FMA per SIMD at 2.4 GHz\n", ILP, waves_per_simd, 2 * fmas / ms / 1e9,
           ms * 1e-3 * 2.4e9 / ((double)iters * ILP * waves_per_simd));
}
// One or two dependent FP64 chains with F independent FP32 FMAs (on other registers) between consecutive FP64 instructions: does any
// instruction fill the dependent-FP64 gap, or only FP64 ones?
template <int ILP, int F>
__global__ __launch_bounds__(256) void kmix(double* out, int iters, double a, double b, float fa, float fb) {
    double x[ILP];
    float y[4];
#pragma unroll
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            x[i] = __builtin_fma(x[i], a, b);
#pragma unroll
            for (int f = 0; f < F; ++f) y[(i * F + f) & 3] = __builtin_fmaf(y[(i * F + f) & 3], fa, fb);
            asm volatile("" ::: "memory");
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += y[i];
    if (s == 12345.678) out[0] = s;
}
template <int ILP, int F>
void runmix(double* d, int waves_per_simd) {
    const int iters = 20000;
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kmix<ILP, F>), dim3(blocks), dim3(256), 0, 0, d, 100, 1.0000001, 1e-9, 1.0000001f, 1e-9f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kmix<ILP, F>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9, 1.0000001f, 1e-9f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mixed: %d FP64 chain(s) + %d FP32 FMA(s) after each FP64, %d waves/SIMD: %.2f cycles per FP64 instruction per wave per SIMD, %.2f per instruction of either kind\n", ILP, F,
           waves_per_simd, ms * 1e-3 * 2.4e9 / ((double)iters * ILP * waves_per_simd), ms * 1e-3 * 2.4e9 / ((double)iters * ILP * (1 + F) * waves_per_simd));
}
int main() {
    double* d; hipMalloc(&d, 64);
    for (int w : {1, 2, 4, 8}) { run<1>(d, w); run<2>(d, w); run<3>(d, w); run<4>(d, w); run<6>(d, w); run<8>(d, w); }
    for (int w : {2, 4}) { runmix<1, 1>(d, w); runmix<1, 2>(d, w); runmix<1, 3>(d, w); runmix<2, 2>(d, w); }
    return 0;
}
