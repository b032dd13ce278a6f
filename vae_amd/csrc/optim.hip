// Global-norm gradient clipping + AdamW over the flat parameter / gradient buffers:
//   torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)   vae/training/train.py:336
//   torch.optim.AdamW(model.parameters(), lr).step()                   vae/training/train.py:310,338
// Two launches for the whole model instead of ~10 multi-tensor launches over 52 tensors:
// (1) per-workgroup sums of squares of the gradients, (2) every workgroup folds those partials
// into the clip coefficient and applies the decoupled-weight-decay Adam update to its slice.
// HBM-bound: 28 algorithmic bytes per parameter (g, p, m, v read; p, m, v written).
#include "common.h"
#include "elbo.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

// With e.out set the launch has one workgroup more, which finishes the step's loss sums instead (they have no reader on
// the device: taking them along here saves the step a launch of its own).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partial, int nsum,
                                                    const ElboCombine e) {
    if ((int)blockIdx.x >= nsum) {
        __shared__ double redd[4];
        elbo_combine_body(e, redd);
        return;
    }
    __shared__ float red[8];
    float acc = 0.f;
    const long n4 = n >> 2;
    const float4* g4 = (const float4*)g;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)nsum * blockDim.x) {
        const float4 v = g4[i];
        acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) acc += g[i] * g[i];
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void adamw_clip_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, long n,
                                                         const float* __restrict__ partial, int npartial, float grad_mul,
                                                         float max_norm, float lr, float b1, float b2, float eps,
                                                         float wd, float bc1, float bc2_sqrt, float omb1, float omb2,
                                                         float* __restrict__ norm_out) {
    __shared__ double red[4];
    __shared__ float coef_s;
    double a = 0.0;
    for (int i = threadIdx.x; i < npartial; i += blockDim.x) a += (double)partial[i];
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0] + red[1] + red[2] + red[3]) * grad_mul;   // norm of grad_mul * g
        float c = max_norm > 0.f ? max_norm / (norm + 1e-6f) : 1.f;
        coef_s = (c < 1.f ? c : 1.f) * grad_mul;
        if (blockIdx.x == 0 && norm_out) *norm_out = norm;
    }
    __syncthreads();
    const float coef = coef_s;
    const float step = lr, decay = wd;   // the host passes lr / bias_correction1 and 1 - lr * weight_decay
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 P = ((float4*)p)[i], M = ((float4*)m)[i], V = ((float4*)v)[i];
        const float4 G = ((const float4*)g)[i];
        float pp[4] = {P.x, P.y, P.z, P.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
        const float gg[4] = {G.x, G.y, G.z, G.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * coef;
            pp[k] *= decay;
            mm[k] = b1 * mm[k] + omb1 * gk;
            vv[k] = b2 * vv[k] + omb2 * gk * gk;
            pp[k] -= step * mm[k] / (sqrtf(vv[k]) / bc2_sqrt + eps);
        }
        ((float4*)p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        ((float4*)m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        ((float4*)v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
            const float gk = g[i] * coef;
            float pk = p[i] * decay;
            const float mk = b1 * m[i] + omb1 * gk, vk = b2 * v[i] + omb2 * gk * gk;
            pk -= step * mk / (sqrtf(vk) / bc2_sqrt + eps);
            p[i] = pk; m[i] = mk; v[i] = vk;
        }
}

}  // namespace vae

using namespace vae;

extern "C" int vae_adamw_grid(long n) {
    long b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

extern "C" int vae_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                   float* partial, float grad_mul, float max_norm, double lr, double beta1, double beta2,
                                   double eps, double weight_decay, int step, float* grad_norm_out, void* stream) {
    return vae_adamw_clip_step_elbo(params, grads, exp_avg, exp_avg_sq, n, partial, grad_mul, max_norm, lr, beta1, beta2, eps,
                                    weight_decay, step, grad_norm_out, nullptr, 0, nullptr, 0, 0.f, 0.f, nullptr, stream);
}

extern "C" int vae_adamw_clip_step_elbo(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                                        float* partial, float grad_mul, float max_norm, double lr, double beta1, double beta2,
                                        double eps, double weight_decay, int step, float* grad_norm_out,
                                        const float* recon_partial, int n_recon, const float* kl_partial, int n_kl,
                                        float kl_weight, float inv_batch, float* out3, void* stream) {
    VAE_CHECK(params && grads && exp_avg && exp_avg_sq && partial, "vae_adamw_clip_step: null pointer");
    VAE_CHECK(!out3 || ((recon_partial || n_recon == 0) && (kl_partial || n_kl == 0) && n_recon >= 0 && n_kl >= 0),
              "vae_adamw_clip_step_elbo: null partial sums");
    VAE_CHECK(n > 0 && step >= 1, "vae_adamw_clip_step: n=%ld, step=%d must be positive", n, step);
    VAE_CHECK((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
              "vae_adamw_clip_step: buffers must be 16-byte aligned");
    const int grid = vae_adamw_grid(n);
    hipStream_t s = (hipStream_t)stream;
    // hyper-parameter arithmetic in double, as torch does on the host, then rounded once
    const float bc1 = (float)(1.0 - pow(beta1, (double)step));
    const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)step));
    {
        ProfScope prof("sumsq_kernel", 0.0, 4.0 * n, s);
        const ElboCombine e = {recon_partial, kl_partial, out3, n_recon, n_kl, kl_weight, inv_batch};
        hipLaunchKernelGGL(sumsq_kernel, dim3(grid + (out3 ? 1 : 0)), dim3(256), 0, s, grads, n, partial, grid, e);
    }
    VAE_LAUNCH_CHECK("sumsq_kernel");
    {
        ProfScope prof("adamw_clip_kernel", 0.0, 28.0 * n, s);
        hipLaunchKernelGGL(adamw_clip_kernel, dim3(grid), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n, partial,
                           grid, grad_mul, max_norm, (float)(lr / (1.0 - pow(beta1, (double)step))), (float)beta1,
                           (float)beta2, (float)eps, (float)(1.0 - lr * weight_decay), bc1, bc2s, (float)(1.0 - beta1),
                           (float)(1.0 - beta2), grad_norm_out);
    }
    VAE_LAUNCH_CHECK("adamw_clip_kernel");
    return 0;
}
