// Microbenchmark: folding the per-workgroup partial rows of a launch INSIDE that launch (the last H workgroups to
// finish wait for the stragglers and fold a slice of the columns each) against the separate fold launch it would
// replace.  Rows are written through (sc1 stores), arrival is one relaxed agent-scope ticket per workgroup, helpers
// poll the ticket counter with sc1 loads and read the rows behind an agent-scope acquire.
//   hipcc --offload-arch=gfx950 -O3 -o tools/tail_fold tools/tail_fold.hip && tools/tail_fold
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float work(int spin) {
    float v = threadIdx.x * 0.001f + blockIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    return v;
}

// plain: rows only (the fold is another launch)
__global__ __launch_bounds__(512) void produce_kernel(float* rows, int ncols, int spin) {
    const float v = work(spin);
    for (int j = threadIdx.x; j < ncols; j += blockDim.x) rows[(long)blockIdx.x * ncols + j] = v + j;
}
__global__ __launch_bounds__(1024) void fold_kernel(const float* rows, int nrows, int ncols, float* out) {
    __shared__ double red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double s = 0.0;
    if (c < ncols) for (int r = rg; r < nrows; r += 32) s += (double)rows[(long)r * ncols + c];
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < ncols) { double t = 0.0; for (int i = 0; i < 32; ++i) t += red[i][cl]; out[c] = (float)t; }
}

// fused: the last H arrivers fold.  `epoch` makes the counter monotonic (no reset between launches).
__global__ __launch_bounds__(512) void produce_fold_kernel(float* rows, int ncols, int spin, unsigned long long* counter,
                                                           unsigned long long epoch, int H, float* out, int* timeouts) {
    __shared__ unsigned long long ticket_s;
    __shared__ double red[16][33];
    const float v = work(spin);
    for (int j = threadIdx.x; j < ncols; j += blockDim.x)
        __hip_atomic_store(rows + (long)blockIdx.x * ncols + j, v + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: written through
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long T = gridDim.x, base = epoch * T;
    if (threadIdx.x == 0) ticket_s = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base;
    __syncthreads();
    const unsigned long long ticket = ticket_s;
    if (ticket + H < T) return;
    const int h = (int)(ticket + H - T);            // helper 0 .. H-1
    if (threadIdx.x == 0) {
        int polls = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < base + T) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1 << 22)) { atomicAdd(timeouts, 1); break; }     // never hang
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // this helper's columns: slices of 32
    const int nslice = (ncols + 31) / 32;
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;      // 16 row groups
    for (int sl = h; sl < nslice; sl += H) {
        const int c = sl * 32 + cl;
        double s = 0.0;
        if (c < ncols)
            for (int r = rg; r < (int)T; r += 16)
                s += (double)__hip_atomic_load(rows + (long)r * ncols + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[rg][cl] = s;
        __syncthreads();
        if (rg == 0 && c < ncols) { double t = 0.0; for (int i = 0; i < 16; ++i) t += red[i][cl]; out[c] = (float)t; }
        __syncthreads();
    }
}

int main() {
    float *rows, *out, *out2; unsigned long long* counter; int* timeouts;
    CK(hipMalloc(&rows, 1024L * 1024 * 4)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&out2, 4096));
    CK(hipMalloc(&counter, 8)); CK(hipMalloc(&timeouts, 4));
    CK(hipMemset(timeouts, 0, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    printf("{\"us_per_launch\": [\n");
    bool first = true;
    for (int nwg : {256, 512, 768})
        for (int ncols : {128, 256, 512})
            for (int spin : {2000, 20000}) {
                const int iters = 200;
                for (int i = 0; i < 5; ++i) {
                    hipLaunchKernelGGL(produce_kernel, dim3(nwg), dim3(512), 0, 0, rows, ncols, spin);
                    hipLaunchKernelGGL(fold_kernel, dim3((ncols + 31) / 32), dim3(1024), 0, 0, rows, nwg, ncols, out);
                }
                CK(hipEventRecord(a));
                for (int i = 0; i < iters; ++i) {
                    hipLaunchKernelGGL(produce_kernel, dim3(nwg), dim3(512), 0, 0, rows, ncols, spin);
                    hipLaunchKernelGGL(fold_kernel, dim3((ncols + 31) / 32), dim3(1024), 0, 0, rows, nwg, ncols, out);
                }
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms0; CK(hipEventElapsedTime(&ms0, a, b));
                CK(hipEventRecord(a));
                for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(produce_kernel, dim3(nwg), dim3(512), 0, 0, rows, ncols, spin);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float msp; CK(hipEventElapsedTime(&msp, a, b));
                float best[3];
                int hs[3] = {8, 32, 64};
                for (int k = 0; k < 3; ++k) {
                    CK(hipMemset(counter, 0, 8));
                    unsigned long long epoch = 0;
                    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(produce_fold_kernel, dim3(nwg), dim3(512), 0, 0, rows, ncols, spin, counter, epoch++, hs[k], out2, timeouts);
                    CK(hipEventRecord(a));
                    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(produce_fold_kernel, dim3(nwg), dim3(512), 0, 0, rows, ncols, spin, counter, epoch++, hs[k], out2, timeouts);
                    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                    CK(hipEventElapsedTime(&best[k], a, b));
                }
                std::vector<float> h1(ncols), h2(ncols);
                CK(hipMemcpy(h1.data(), out, ncols * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), out2, ncols * 4, hipMemcpyDeviceToHost));
                int diff = 0; for (int j = 0; j < ncols; ++j) diff += h1[j] != h2[j];
                printf("%s {\"workgroups\": %d, \"cols\": %d, \"work_spin\": %d, \"producer_alone\": %.2f, \"producer_plus_fold_launch\": %.2f, "
                       "\"fused_8_helpers\": %.2f, \"fused_32_helpers\": %.2f, \"fused_64_helpers\": %.2f, \"mismatching_columns\": %d}",
                       first ? "" : ",\n", nwg, ncols, spin, msp * 1000 / iters, ms0 * 1000 / iters, best[0] * 1000 / iters, best[1] * 1000 / iters,
                       best[2] * 1000 / iters, diff);
                first = false;
            }
    int to; CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
    printf("\n], \"poll_timeouts\": %d}\n", to);
    return 0;
}
