// nn.Conv2d(64, Cout, 1) on its own: the output layer of the drafts' U-Net models, which return its logits
// (vae/practice/vae_original.py:119,173-175 -- no normalisation in front and no sigmoid behind, unlike the live model's
// head, vae/models/vae.py:74,122, which norm.hip fuses with bn1 + ReLU + sigmoid).  HBM-bound: the 64-channel NHWC bf16
// tensor is read once, Cout <= 8 planes of fp32 NCHW leave.  8 lanes share one pixel (8 channels each).
//   forward : out[n][o][h][w] = sum_c x[n][h][w][c] w[o][c] + b[o]
//   backward: dx[pix][c] = sum_o g[o][pix] w[o][c];  dW[o][c] = sum_pix g[o][pix] x[pix][c];  db[o] = sum_pix g[o][pix]
//             (one partial row of (Cout+1)*64 floats per workgroup, folded in a fixed order by vae_sum_partials)
#include "common.h"
#include "act.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

template <int CO>
__global__ __launch_bounds__(256) void conv1x1_out_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, float* __restrict__ out,
                                                              long npix, int hw) {
    const int sub = threadIdx.x & 7, c0 = sub * 8;
    float wv[CO][8];
#pragma unroll
    for (int o = 0; o < CO; ++o) load8f(w + o * 64 + c0, wv[o]);
    const long stride = (long)gridDim.x * 32;
    const long iters = (npix + stride - 1) / stride;
    for (long it = 0; it < iters; ++it) {        // every lane takes every trip: the shuffles below need the whole group
        const long pix = it * stride + (long)blockIdx.x * 32 + (threadIdx.x >> 3);
        const bool ok = pix < npix;
        float v[8], acc[CO];
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[o] = 0.f;
        if (ok) {
            load8(x + pix * 64 + c0, v);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int o = 0; o < CO; ++o) acc[o] += v[i] * wv[o][i];
        }
#pragma unroll
        for (int o = 0; o < CO; ++o) {
            acc[o] += __shfl_xor(acc[o], 1, 64);
            acc[o] += __shfl_xor(acc[o], 2, 64);
            acc[o] += __shfl_xor(acc[o], 4, 64);
        }
        if (ok && sub < CO) {
            float l = 0.f;
#pragma unroll
            for (int o = 0; o < CO; ++o) if (sub == o) l = acc[o] + b[o];
            const long img = pix / hw, p = pix - img * hw;
            out[(img * CO + sub) * hw + p] = l;
        }
    }
}

template <int CO>
__global__ __launch_bounds__(256) void conv1x1_out_bwd_kernel(const float* __restrict__ g, const bf16_t* __restrict__ x,
                                                              const float* __restrict__ w, float* __restrict__ partial,
                                                              bf16_t* __restrict__ dx, long npix, int hw) {
    constexpr int ROWS = CO + 1;
    __shared__ float red[ROWS][32][64 + 1];
    const int sub = threadIdx.x & 7, c0 = sub * 8, pl = threadIdx.x >> 3;
    float wv[CO][8], a_w[CO][8], a_b[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        load8f(w + o * 64 + c0, wv[o]);
        a_b[o] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) a_w[o][i] = 0.f;
    }
    const long stride = (long)gridDim.x * 32;
    for (long pix = (long)blockIdx.x * 32 + pl; pix < npix; pix += stride) {
        const long img = pix / hw, p = pix - img * hw;
        float dl[CO], v[8], o8[8];
#pragma unroll
        for (int o = 0; o < CO; ++o) dl[o] = g[(img * CO + o) * hw + p];
        load8(x + pix * 64 + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float d = 0.f;
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                d += dl[o] * wv[o][i];
                a_w[o][i] += dl[o] * v[i];
            }
            o8[i] = d;
        }
        if (sub == 0) {
#pragma unroll
            for (int o = 0; o < CO; ++o) a_b[o] += dl[o];
        }
        if (dx) store8(dx + pix * 64 + c0, o8);
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[o][pl][c0 + i] = a_w[o][i];
    }
    if (sub == 0) {
#pragma unroll
        for (int o = 0; o < CO; ++o) red[CO][pl][o] = a_b[o];
    }
    __syncthreads();
    for (int j = threadIdx.x; j < ROWS * 64; j += blockDim.x) {
        const int row = j >> 6, c = j & 63;
        float s = 0.f;
        if (row < CO || c < CO)
            for (int t = 0; t < 32; ++t) s += red[row][t][c];
        partial[(long)blockIdx.x * ROWS * 64 + j] = s;
    }
}

static int out_grid(long npix) {
    long b = (npix + 31) / 32;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

}  // namespace vae

using namespace vae;

extern "C" int vae_conv1x1_out_fwd(const void* x, const float* w, const float* b, float* out, int n_img, int H, int W,
                                   int Cout, void* stream) {
    VAE_CHECK(x && w && b && out, "vae_conv1x1_out_fwd: null pointer");
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31), "vae_conv1x1_out_fwd: bad geometry");
    VAE_CHECK(Cout == 1 || Cout == 3 || Cout == 4, "vae_conv1x1_out_fwd: Cout=%d (built for 1, 3, 4 output channels)", Cout);
    const long npix = (long)n_img * H * W;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("conv1x1_out_fwd_kernel", 2.0 * npix * 64 * Cout, npix * (128.0 + 4.0 * Cout), s);
#define GO(CO) hipLaunchKernelGGL(conv1x1_out_fwd_kernel<CO>, dim3(out_grid(npix)), dim3(256), 0, s, (const bf16_t*)x, w, b, out, npix, H * W)
    if (Cout == 1) GO(1); else if (Cout == 3) GO(3); else GO(4);
#undef GO
    VAE_LAUNCH_CHECK("conv1x1_out_fwd_kernel");
    return 0;
}

extern "C" int vae_conv1x1_out_bwd_blocks(int n_img, int H, int W) {
    long b = ((long)n_img * H * W + 32 * 16 - 1) / (32 * 16);      // >= 16 pixels per lane group before a second workgroup
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

extern "C" int vae_conv1x1_out_bwd(const float* gout, const void* x, const float* w, float* partial, float* sums,
                                   void* dx, int n_img, int H, int W, int Cout, void* stream) {
    VAE_CHECK(gout && x && w && partial && sums, "vae_conv1x1_out_bwd: null pointer");
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31), "vae_conv1x1_out_bwd: bad geometry");
    VAE_CHECK(Cout == 1 || Cout == 3 || Cout == 4, "vae_conv1x1_out_bwd: Cout=%d (built for 1, 3, 4 output channels)", Cout);
    const long npix = (long)n_img * H * W;
    const int nb = vae_conv1x1_out_bwd_blocks(n_img, H, W);
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope prof("conv1x1_out_bwd_kernel", 4.0 * npix * 64 * Cout, npix * (256.0 + 4.0 * Cout), s);
#define GO(CO) hipLaunchKernelGGL(conv1x1_out_bwd_kernel<CO>, dim3(nb), dim3(256), 0, s, gout, (const bf16_t*)x, w, partial, (bf16_t*)dx, npix, H * W)
        if (Cout == 1) GO(1); else if (Cout == 3) GO(3); else GO(4);
#undef GO
        VAE_LAUNCH_CHECK("conv1x1_out_bwd_kernel");
    }
    return vae_sum_partials(partial, nb, (Cout + 1) * 64, sums, 1.0f, stream);
}
