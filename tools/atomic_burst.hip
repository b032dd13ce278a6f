// Microbenchmark: what does it cost a launch when every workgroup ends by adding 2*C 64-bit integers into the SAME
// 2*C accumulators (order-independent sums), against leaving a row of floats per workgroup?
//   hipcc --offload-arch=gfx950 -O3 -o tools/atomic_burst tools/atomic_burst.hip && tools/atomic_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: plain row store; 1: int64 atomics into one set; 2: into 8 replicas (by workgroup)
template <int MODE>
__global__ __launch_bounds__(256) void tail_kernel(float* rows, unsigned long long* acc, int ncols, int spin) {
    // a little work in front so that workgroups do not all arrive in the same cycle
    float v = threadIdx.x * 0.001f + blockIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    for (int j = threadIdx.x; j < ncols; j += 256) {
        if (MODE == 0) rows[(long)blockIdx.x * ncols + j] = v;
        else {
            const long long q = (long long)(v * 1024.f);
            const int rep = MODE == 2 ? (blockIdx.x & 7) : 0;
            __hip_atomic_fetch_add(acc + (long)rep * ncols + j, (unsigned long long)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// consumer: every workgroup folds `nbins` accumulators per column through LDS before its own work
__global__ __launch_bounds__(256) void consumer_kernel(const unsigned long long* acc, int ncols, int nbins, float* out, int fold) {
    __shared__ float tot[1024];
    if (fold) {
        for (int j = threadIdx.x; j < ncols; j += 256) {
            double s = 0.0;
            for (int b = 0; b < nbins; ++b) s += (double)(long long)acc[(long)b * ncols + j];
            tot[j] = (float)s;
        }
        __syncthreads();
    }
    float r = fold ? tot[threadIdx.x % ncols] : (float)threadIdx.x;
    out[(long)blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
float run_tail(float* rows, unsigned long long* acc, int nwg, int ncols, int spin, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(tail_kernel<MODE>, dim3(nwg), dim3(256), 0, 0, rows, acc, ncols, spin);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(tail_kernel<MODE>, dim3(nwg), dim3(256), 0, 0, rows, acc, ncols, spin);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / iters;
}

int main() {
    float* rows; unsigned long long* acc; float* out;
    CK(hipMalloc(&rows, 4096L * 2048 * 4)); CK(hipMalloc(&acc, 8 * 3 * 2048 * 8)); CK(hipMalloc(&out, 32768L * 256 * 4));
    CK(hipMemset(acc, 0, 8 * 3 * 2048 * 8));
    printf("{\"tail_us_per_launch\": [\n");
    const int nwgs[] = {256, 512, 768, 1024, 4096};
    const int cols[] = {128, 256, 512};
    bool first = true;
    for (int spin : {0, 2000})
        for (int nwg : nwgs)
            for (int nc : cols) {
                const float t0 = run_tail<0>(rows, acc, nwg, nc, spin, 200);
                const float t1 = run_tail<1>(rows, acc, nwg, nc, spin, 200);
                const float t2 = run_tail<2>(rows, acc, nwg, nc, spin, 200);
                printf("%s {\"spin\": %d, \"workgroups\": %d, \"cols\": %d, \"row_store\": %.2f, \"atomic_one_set\": %.2f, \"atomic_8_sets\": %.2f}", first ? "" : ",\n", spin, nwg, nc, t0, t1, t2);
                first = false;
            }
    printf("\n], \"consumer_us_per_launch\": [\n");
    first = true;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nwg : {2048, 16384})
        for (int nc : {128, 512})
            for (int fold : {0, 1, 3}) {
                for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(consumer_kernel, dim3(nwg), dim3(256), 0, 0, acc, nc, fold ? fold : 1, out, fold ? 1 : 0);
                hipEventRecord(a);
                for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(consumer_kernel, dim3(nwg), dim3(256), 0, 0, acc, nc, fold ? fold : 1, out, fold ? 1 : 0);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("%s {\"workgroups\": %d, \"cols\": %d, \"bins_folded\": %d, \"us\": %.2f}", first ? "" : ",\n", nwg, nc, fold, ms * 5.f);
                first = false;
            }
    printf("\n]}\n");
    return 0;
}
