// loc_lib_amd/csrc/bfnn.hip — BfnnRegistration (LocUtils/src/model/search_point/bfnn/bfnn.cpp:14-50): brute-force k nearest points.
//
// The reference computes every float32 squared distance ((p − q).squaredNorm(), Eigen's x0 + (x1 + x2) order), std::sorts the whole
// (index, distance) array by distance and returns the first k indices. The order std::sort leaves among EQUAL distances is
// unspecified; here equal distances are ordered by point index. One 256-thread workgroup per query: every thread keeps the k best of
// its strided share of the cloud in registers (coalesced 16-byte loads, the cloud streams through L2 once per resident query wave),
// then k rounds of a block-wide arg-min pick the winners.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "context.hpp"

namespace locgpu {

constexpr int kBfBlock = 256;
constexpr int kBfMaxK = 8;

template <int K>
__global__ __launch_bounds__(kBfBlock) void bfnn_kernel(const float4* __restrict__ pts, uint32_t n, const float* __restrict__ queries, int k,
                                                        int32_t* __restrict__ out) {
    __shared__ float s_d[kBfBlock / 64];
    __shared__ uint32_t s_i[kBfBlock / 64];
    __shared__ uint32_t s_win;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float qx = queries[3 * q], qy = queries[3 * q + 1], qz = queries[3 * q + 2];
    float d[K];
    uint32_t id[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { d[j] = __builtin_inff(); id[j] = 0xFFFFFFFFu; }
    for (uint32_t i = tid; i < n; i += kBfBlock) {
        const float4 p = pts[i];
        const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
        float c = dx * dx + (dy * dy + dz * dz);
        uint32_t ci = i;
        if (c < d[K - 1]) {  // ascending insertion; a thread sees its indices in increasing order, so ties keep the smaller index first
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const bool sm = c < d[j];
                const float nd = sm ? c : d[j], nc = sm ? d[j] : c;
                const uint32_t ni = sm ? ci : id[j], nci = sm ? id[j] : ci;
                d[j] = nd; id[j] = ni; c = nc; ci = nci;
            }
        }
    }
    for (int r = 0; r < k; ++r) {
        // block-wide arg-min over every thread's current head (d[0], id[0]); ties → smaller index
        float bd = d[0];
        uint32_t bi = id[0];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd, off, 64);
            const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off, 64);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        if (lane == 0) { s_d[wave] = bd; s_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float wd = s_d[0];
            uint32_t wi = s_i[0];
            for (int w = 1; w < kBfBlock / 64; ++w)
                if (s_d[w] < wd || (s_d[w] == wd && s_i[w] < wi)) { wd = s_d[w]; wi = s_i[w]; }
            s_win = wi;
            out[(size_t)q * k + r] = wi == 0xFFFFFFFFu ? -1 : (int32_t)wi;
        }
        __syncthreads();
        if (id[0] == s_win && s_win != 0xFFFFFFFFu) {  // the winner pops its head
#pragma unroll
            for (int j = 0; j + 1 < K; ++j) { d[j] = d[j + 1]; id[j] = id[j + 1]; }
            d[K - 1] = __builtin_inff(); id[K - 1] = 0xFFFFFFFFu;
        }
        __syncthreads();
    }
}

}  // namespace locgpu

using namespace locgpu;

extern "C" {

int locgpu_bfnn_set_target(locgpu_ctx* ctx, const void* pts, size_t n, size_t stride_bytes) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!pts || n == 0 || stride_bytes < 12 || n > 0x7FFFFFF0ull) return fail(ctx, LOCGPU_ERR_INVALID, "bfnn_set_target: empty cloud or stride < 12");  // bfnn.cpp:16-19
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<float4> host(n);
    const char* base = (const char*)pts;
    for (size_t i = 0; i < n; ++i) { host[i] = float4{0.f, 0.f, 0.f, 0.f}; std::memcpy(&host[i], base + i * stride_bytes, 12); }
    if (ctx->d_bfnn) { LOCGPU_HIP(ctx, hipFree(ctx->d_bfnn)); ctx->d_bfnn = nullptr; ctx->bfnn_n = 0; }
    LOCGPU_HIP(ctx, hipMalloc((void**)&ctx->d_bfnn, n * sizeof(float4)));
    LOCGPU_HIP(ctx, hipMemcpy(ctx->d_bfnn, host.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    ctx->bfnn_n = n;
    return LOCGPU_OK;
}

int locgpu_bfnn_knn(locgpu_ctx* ctx, const float* queries, size_t nq, int k, int32_t* out_idx) {
    if (!ctx) return LOCGPU_ERR_INVALID;
    if (!ctx->d_bfnn) return fail(ctx, LOCGPU_ERR_NO_TARGET, "bfnn_knn: no target set");
    if (!queries || !out_idx || k < 1 || k > kBfMaxK) return fail(ctx, LOCGPU_ERR_INVALID, "bfnn_knn: bad arguments (1 <= k <= 8)");
    if ((size_t)k > ctx->bfnn_n) return fail(ctx, LOCGPU_ERR_K_TOO_LARGE, "bfnn_knn: k larger than the cloud (the reference reads past the end here)");
    if (nq == 0) return LOCGPU_OK;
    LOCGPU_HIP(ctx, hipSetDevice(ctx->device));
    float* d_q = nullptr;
    int32_t* d_out = nullptr;
    int rc = LOCGPU_OK;
    if (!hip_ok(ctx, hipMalloc((void**)&d_q, nq * 12), "hipMalloc") || !hip_ok(ctx, hipMalloc((void**)&d_out, nq * k * sizeof(int32_t)), "hipMalloc")) rc = LOCGPU_ERR_OOM;
    hipStream_t s = ctx->stream;
    if (rc == LOCGPU_OK && !hip_ok(ctx, hipMemcpyAsync(d_q, queries, nq * 12, hipMemcpyHostToDevice, s), "H2D")) rc = LOCGPU_ERR_NO_DEVICE;
    if (rc == LOCGPU_OK) {
        if (k == 1) hipLaunchKernelGGL((bfnn_kernel<1>), dim3((unsigned)nq), dim3(kBfBlock), 0, s, ctx->d_bfnn, (uint32_t)ctx->bfnn_n, d_q, k, d_out);
        else if (k <= 5) hipLaunchKernelGGL((bfnn_kernel<5>), dim3((unsigned)nq), dim3(kBfBlock), 0, s, ctx->d_bfnn, (uint32_t)ctx->bfnn_n, d_q, k, d_out);
        else hipLaunchKernelGGL((bfnn_kernel<8>), dim3((unsigned)nq), dim3(kBfBlock), 0, s, ctx->d_bfnn, (uint32_t)ctx->bfnn_n, d_q, k, d_out);
        if (!hip_ok(ctx, hipGetLastError(), "bfnn launch") || !hip_ok(ctx, hipMemcpyAsync(out_idx, d_out, nq * k * sizeof(int32_t), hipMemcpyDeviceToHost, s), "D2H") ||
            !hip_ok(ctx, hipStreamSynchronize(s), "sync"))
            rc = LOCGPU_ERR_NO_DEVICE;
    }
    if (d_q) (void)hipFree(d_q);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
