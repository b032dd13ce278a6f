// Microbenchmark: how fast can the chip move global memory into LDS?
//   (a) buffer_load ... lds (the copy engine path: 16 bytes per lane, 1 KB per wave-instruction, no registers)
//   (b) global_load_dwordx4 into registers + ds_write_b128
// from a per-workgroup region that is L2-resident (64 KB, re-read) or streamed (a 1 GiB buffer, read once).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_dma_rate tools/lds_dma_rate.hip && tools/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lptr_t;

// MODE 0: LDS-DMA, MODE 1: registers + ds_write.  Each wave moves `iters` x 4 KB; DEPTH copies of 1 KB in flight per wave.
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 1) void copy_kernel(const unsigned char* src, long region, long wg_stride, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[8 * 4 * 1024 * 4];     // 8 waves x 4 slots x 4 KB
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6;
    const unsigned char* base = src + (long)blockIdx.x * wg_stride;
    unsigned char* my = smem + wave * 16384;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
    float acc = 0.f;
    unsigned off = (unsigned)(wave * 4096 + lane * 16);
    for (int it = 0; it < iters; ++it) {
        unsigned char* dst = my + (it & 3) * 4096;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(dst + k * 1024), 16, (int)(off + k * 1024), 0, 0, 0);
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            uint4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const uint4*)(base + off + k * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k) *(uint4*)(dst + k * 1024 + lane * 16) = v[k];
        }
        off += (unsigned)(nw * 4096);
        if (off + 4096 > (unsigned)region) off = (unsigned)(wave * 4096 + lane * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += *(float*)(smem + threadIdx.x * 4);
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE, int DEPTH>
double run(const unsigned char* src, long region, long wg_stride, int waves, int iters, float* sink) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((copy_kernel<MODE, DEPTH>), dim3(256), dim3(64 * waves), 0, 0, src, region, wg_stride, iters, sink);
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((copy_kernel<MODE, DEPTH>), dim3(256), dim3(64 * waves), 0, 0, src, region, wg_stride, iters, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return 256.0 * waves * iters * 4096.0 * 5 / (ms * 1e-3) / 1e12;       // TB/s
}

int main() {
    unsigned char* src; float* sink;
    const long big = 1L << 30;
    if (hipMalloc(&src, big) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(src, 1, big);
    printf("{\"unit\": \"TB/s into LDS, 256 workgroups\", \"rows\": [\n");
    bool first = true;
    for (int waves : {4, 8})
        for (int resident : {1, 0}) {
            const long region = resident ? 65536 : big / 256;            // per-workgroup region
            const long stride = region;
            const int iters = resident ? 400 : (int)(region / (waves * 4096));
            const double d1 = run<0, 1>(src, region, stride, waves, iters, sink);
            const double d2 = run<0, 2>(src, region, stride, waves, iters, sink);
            const double d4 = run<0, 4>(src, region, stride, waves, iters, sink);
            const double rg = run<1, 1>(src, region, stride, waves, iters, sink);
            printf("%s {\"waves_per_cu\": %d, \"source\": \"%s\", \"lds_dma_4KB_in_flight\": %.2f, \"lds_dma_8KB\": %.2f, \"lds_dma_16KB\": %.2f, \"registers_then_ds_write\": %.2f}",
                   first ? "" : ",\n", waves, resident ? "64 KB per workgroup, L2-resident" : "4 MB per workgroup, streamed once", d1, d2, d4, rg);
            first = false;
        }
    printf("\n]}\n");
    return 0;
}
