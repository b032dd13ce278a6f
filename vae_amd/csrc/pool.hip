// nn.MaxPool2d(2) on its own (vae/models/vae.py:35,38,41; the drafts' U-Net keeps the unpooled tensor for the skip
// connection, vae/practice/vae_original.py:142-149): NHWC bf16, one thread owns 8 consecutive channels of one pooled
// position.  The model's own pooling is fused into the BatchNorm / GroupNorm activation kernels; these entry points
// serve callers that need the pooled and the unpooled tensor both.
//   forward : out = max over the 2x2 window, first maximum in scan order wins (as MaxPool2d); optional route byte
//             per pooled element = window position of the maximum
//   backward: dx = g at the maximum's position, 0 at the other three (from the route bytes, or recomputed from x)
#include "common.h"
#include "act.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

struct PoolGeom {
    int n_img, H, W, C;
};

__device__ __forceinline__ void pool_window(const PoolGeom& g, long item, int cgs, int& c0, long pix[4], long& opos) {
    const int cg = (int)(item % cgs);
    opos = item / cgs;
    const int wo_n = g.W >> 1, ho_n = g.H >> 1;
    const int wo = (int)(opos % wo_n);
    const long t = opos / wo_n;
    const int ho = (int)(t % ho_n);
    const long img = t / ho_n;
    c0 = cg * 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) pix[k] = (img * g.H + 2 * ho + (k >> 1)) * g.W + 2 * wo + (k & 1);
}

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                                                           uint8_t* __restrict__ route, PoolGeom g, long items) {
    const int cgs = g.C >> 3;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        int c0;
        long pix[4], opos;
        pool_window(g, it, cgs, c0, pix, opos);
        float v[4][8], o[8];
        uint32_t arg[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) load8(x + pix[k] * g.C + c0, v[k]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[i] = v[0][i];
            arg[i] = 0;
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][i] > o[i]) { o[i] = v[k][i]; arg[i] = k; }
        }
        store8(out + opos * g.C + c0, o);
        if (route)
            *(uint2*)(route + opos * g.C + c0) = make_uint2(arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24),
                                                            arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24));
    }
}

template <bool ROUTED>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const bf16_t* __restrict__ gout, const bf16_t* __restrict__ x,
                                                           const uint8_t* __restrict__ route, bf16_t* __restrict__ dx,
                                                           PoolGeom g, long items) {
    const int cgs = g.C >> 3;
    for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < items; it += (long)gridDim.x * 256) {
        int c0;
        long pix[4], opos;
        pool_window(g, it, cgs, c0, pix, opos);
        float gi[8];
        uint32_t arg[8];
        load8(gout + opos * g.C + c0, gi);
        if (ROUTED) {
            const uint2 rr = *(const uint2*)(route + opos * g.C + c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) arg[i] = (((i < 4 ? rr.x : rr.y) >> (8 * (i & 3))) & 3u);
        } else {
            float v[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) load8(x + pix[k] * g.C + c0, v[k]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float best = v[0][i];
                arg[i] = 0;
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (v[k][i] > best) { best = v[k][i]; arg[i] = k; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = arg[i] == (uint32_t)k ? gi[i] : 0.f;
            store8(dx + pix[k] * g.C + c0, o);
        }
    }
}

// column sums of a [rows][C] bf16 matrix (the bias gradient of a conv / transposed conv whose output gradient it is):
// partial[block][C] per workgroup, folded in a fixed order by the caller's second launch
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ x, long rows, int C,
                                                          float* __restrict__ partial) {
    __shared__ float red[256][8];
    const int cgs = C >> 3, cshift = __ffs(cgs) - 1;
    const int c0 = (threadIdx.x & (cgs - 1)) * 8;
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.f;
    const long rstride = ((long)gridDim.x * 256) >> cshift;
    for (long r = ((long)blockIdx.x * 256 + threadIdx.x) >> cshift; r < rows; r += rstride) {
        float v[8];
        load8(x + r * C + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = a[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int t = c >> 3; t < 256; t += cgs) s += red[t][c & 7];
        partial[(long)blockIdx.x * C + c] = s;
    }
}

// torch.clamp(x, lo, hi) and its backward on fp32 (the deeper U-Net draft clamps logvar, vae/practice/vae_original.py:390):
// g == null: out = min(max(x, lo), hi);  g != null: out = g where lo <= x <= hi, else 0
__global__ __launch_bounds__(256) void clamp_f32_kernel(const float* __restrict__ x, const float* __restrict__ g, float lo,
                                                        float hi, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = x[i];
        out[i] = g ? ((v >= lo && v <= hi) ? g[i] : 0.f) : fminf(fmaxf(v, lo), hi);
    }
}

static int pool_grid(long items) {
    long b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

}  // namespace vae

using namespace vae;

extern "C" int vae_maxpool2_fwd(const void* x, void* out, void* route, int n_img, int H, int W, int C, void* stream) {
    VAE_CHECK(x && out, "vae_maxpool2_fwd: null pointer");
    VAE_CHECK(((uintptr_t)route & 7) == 0, "vae_maxpool2_fwd: route must be 8-byte aligned");
    if (int e = check_geom("vae_maxpool2_fwd", n_img, H, W, C, 1)) return e;
    const PoolGeom g = {n_img, H, W, C};
    const long items = (long)n_img * (H / 2) * (W / 2) * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("maxpool2_fwd_kernel", 0.0, 2.0 * n_img * H * W * C * 1.25, s);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(pool_grid(items)), dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)out,
                       (uint8_t*)route, g, items);
    VAE_LAUNCH_CHECK("maxpool2_fwd_kernel");
    return 0;
}

extern "C" int vae_maxpool2_bwd(const void* gout, const void* x, const void* route, void* dx, int n_img, int H, int W, int C,
                                void* stream) {
    VAE_CHECK(gout && dx && (x || route), "vae_maxpool2_bwd: null pointer (the input or the route bytes are needed)");
    VAE_CHECK(((uintptr_t)route & 7) == 0, "vae_maxpool2_bwd: route must be 8-byte aligned");
    if (int e = check_geom("vae_maxpool2_bwd", n_img, H, W, C, 1)) return e;
    const PoolGeom g = {n_img, H, W, C};
    const long items = (long)n_img * (H / 2) * (W / 2) * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("maxpool2_bwd_kernel", 0.0, 2.0 * n_img * H * W * C * (route ? 1.25 : 2.25), s);
    if (route)
        hipLaunchKernelGGL(maxpool2_bwd_kernel<true>, dim3(pool_grid(items)), dim3(256), 0, s, (const bf16_t*)gout,
                           (const bf16_t*)x, (const uint8_t*)route, (bf16_t*)dx, g, items);
    else
        hipLaunchKernelGGL(maxpool2_bwd_kernel<false>, dim3(pool_grid(items)), dim3(256), 0, s, (const bf16_t*)gout,
                           (const bf16_t*)x, (const uint8_t*)route, (bf16_t*)dx, g, items);
    VAE_LAUNCH_CHECK("maxpool2_bwd_kernel");
    return 0;
}

extern "C" int vae_colsum_blocks(long rows, int C) {
    long b = (rows * (C / 8) + 255) / 256;
    if (b < 1) b = 1;
    if (b > 512) b = 512;
    return (int)b;
}

extern "C" int vae_colsum_bf16(const void* x, long rows, int C, float* partial, float* out, void* stream) {
    VAE_CHECK(x && partial && out && rows > 0, "vae_colsum_bf16: bad arguments");
    VAE_CHECK(C > 0 && C % 8 == 0 && 256 % (C / 8) == 0, "vae_colsum_bf16: C=%d must be 8 * a divisor of 256", C);
    const int nb = vae_colsum_blocks(rows, C);
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope prof("colsum_bf16_kernel", 0.0, 2.0 * rows * C, s);
        hipLaunchKernelGGL(colsum_bf16_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, rows, C, partial);
        VAE_LAUNCH_CHECK("colsum_bf16_kernel");
    }
    return vae_sum_partials(partial, nb, C, out, 1.0f, stream);
}

extern "C" int vae_clamp_f32(const float* x, const float* g, float lo, float hi, float* out, long n, void* stream) {
    VAE_CHECK(x && out && n > 0 && lo <= hi, "vae_clamp_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("clamp_f32_kernel", 0.0, (g ? 12.0 : 8.0) * n, s);
    hipLaunchKernelGGL(clamp_f32_kernel, dim3(pool_grid(n)), dim3(256), 0, s, x, g, lo, hi, out, n);
    VAE_LAUNCH_CHECK("clamp_f32_kernel");
    return 0;
}
