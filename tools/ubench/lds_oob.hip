// What does gfx950 do with LDS accesses outside the workgroup's allocation? (search_walk.hpp relies on: reads return 0, writes are
// dropped, and a negative base plus an instruction offset wraps in 32 bits.)   hipcc --offload-arch=gfx950 lds_oob.hip -o lds_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t __attribute__((address_space(3))) lds_u32;
__global__ void k(uint32_t* out, int bytes) {
    extern __shared__ uint32_t s[];
    const int tid = threadIdx.x;
    for (int i = tid; i < bytes / 4; i += blockDim.x) s[i] = 0xA0000000u + i;
    __syncthreads();
    const uint32_t base = (uint32_t)(size_t)s;
    if (tid == 0) {
        out[0] = base;
        out[1] = *reinterpret_cast<lds_u32*>(base - 4u);           // just below
        out[2] = *reinterpret_cast<lds_u32*>(base - 2048u);        // far below
        out[3] = *reinterpret_cast<lds_u32*>(base + bytes);        // first byte beyond
        out[4] = *reinterpret_cast<lds_u32*>(base + bytes + 1276); // within a 1280-byte granule beyond
        out[5] = *reinterpret_cast<lds_u32*>(base + 65536 + 16);   // + 64 KB
        out[6] = *reinterpret_cast<lds_u32*>(base + 163840);       // + 160 KB
        // negative base + instruction offset
        uint32_t neg = base - 1024u + 8u, v;
        asm volatile("ds_read_b32 %0, %1 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(neg));
        out[7] = v;  // expect s[2]
        uint32_t a, b2;
        asm volatile("ds_read2st64_b32 %0, %1 offset0:4 offset1:6\n s_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&out[16]) : "v"(base - 1024u + 4u));
        (void)a; (void)b2;
        // writes beyond and below
        *reinterpret_cast<lds_u32*>(base + bytes) = 0xDEAD0001u;
        *reinterpret_cast<lds_u32*>(base - 4u) = 0xDEAD0002u;
        *reinterpret_cast<lds_u32*>(base + bytes + 512) = 0xDEAD0003u;
    }
    __syncthreads();
    if (tid == 0) {
        out[8] = s[0]; out[9] = s[bytes / 4 - 1];
        out[10] = *reinterpret_cast<lds_u32*>(base + bytes);
        out[11] = *reinterpret_cast<lds_u32*>(base - 4u);
        uint32_t bad = 0;
        for (int i = 0; i < bytes / 4; ++i) bad += s[i] != 0xA0000000u + i;
        out[12] = bad;
    }
}
int main() {
    uint32_t* d; hipMalloc(&d, 256); 
    for (int bytes : {7680, 7168, 30720}) {
        hipMemset(d, 0xFF, 256);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), bytes, 0, d, bytes);
        uint32_t h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("bytes %d: base %u | below4 %08x below2048 %08x beyond0 %08x beyond+1276 %08x +64K %08x +160K %08x | negbase+off %08x (want %08x) | read2st64 %08x %08x (want %08x %08x) | after writes: s[0] %08x s[last] %08x beyond %08x below %08x corrupted %u\n",
               bytes, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], 0xA0000000u + 2, h[16], h[17], 0xA0000000u + 1, 0xA0000000u + 1 + 128, h[8], h[9], h[10], h[11], h[12]);
    }
    return 0;
}
