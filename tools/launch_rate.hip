// Host cost of a kernel launch on this machine: enqueue time per hipLaunchKernelGGL for an empty kernel with a small
// and with a 2 KB by-value argument, queue kept short (sync every 64 launches) and left to run ahead.
//   hipcc -O2 --offload-arch=gfx950 -o tools/launch_rate tools/launch_rate.hip && ./tools/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { int v[512]; };
__global__ void k_small(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b, int* p) { if (p && threadIdx.x == 9999) *p = b.v[3]; }
template <class F> static double per_launch_us(F f, int n, int sync_every) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) { f(); if (sync_every && (i + 1) % sync_every == 0) hipStreamSynchronize(0); }
    auto t1 = std::chrono::steady_clock::now();
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
    Big b = {};
    hipStream_t s; hipStreamCreate(&s);
    for (int w = 0; w < 100; ++w) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, (int*)nullptr);
    hipDeviceSynchronize();
    const double a = per_launch_us([&] { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, (int*)nullptr); }, 20000, 0);
    const double c = per_launch_us([&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, b, (int*)nullptr); }, 20000, 0);
    const double d = per_launch_us([&] { hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, s, (int*)nullptr); (void)hipGetLastError(); }, 20000, 0);
    printf("{\"enqueue_us_small_arg\": %.2f, \"enqueue_us_2KB_arg\": %.2f, \"enqueue_us_256wg_with_error_check\": %.2f}\n", a, c, d);
    return 0;
}
