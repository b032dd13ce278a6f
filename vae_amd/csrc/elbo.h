// The final sums of the ELBO, shared by the stand-alone combine kernel and by the optimizer's first launch (which
// takes them along so that the step has one launch less).
#pragma once
#include "common.h"

namespace vae {

struct ElboCombine {
    const float* recon_partial;
    const float* kl_partial;
    float* out;            // [0] = (recon_sum + kl_weight * kl_sum) * inv_batch, [1] = recon_sum, [2] = kl_sum
    int n_recon, n_kl;
    float kl_weight, inv_batch;
};

// one workgroup of 256 threads; red: 4 doubles of LDS
__device__ __forceinline__ void elbo_combine_body(const ElboCombine& e, double* red) {
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < e.n_recon; i += blockDim.x) a += (double)e.recon_partial[i];
    for (int i = threadIdx.x; i < e.n_kl; i += blockDim.x) b += (double)e.kl_partial[i];
    a = wave_sum_d(a);
    b = wave_sum_d(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = a;
    __syncthreads();
    const double ta = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (lane == 0) red[w] = b;
    __syncthreads();
    const double tb = red[0] + red[1] + red[2] + red[3];
    if (threadIdx.x == 0) {
        e.out[0] = (float)((ta + (double)e.kl_weight * tb) * (double)e.inv_batch);
        e.out[1] = (float)ta;
        e.out[2] = (float)tb;
    }
}

}  // namespace vae
