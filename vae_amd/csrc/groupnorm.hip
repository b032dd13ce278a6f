// GroupNorm + ReLU (+ Dropout) (+ MaxPool2d(2)) forward and backward for the GroupNorm form of the double-conv block
// (reference drafts: vae/practice/vae_conv.py:20-36, vae/practice/vae_original.py:58-74; GroupNorm(8, C), eps 1e-5).
// Activations are NHWC bf16.  GroupNorm statistics belong to ONE image, so one workgroup owns one image: it reads
// the image once for the statistics and a second time (from L2: an image is 8 KB .. 2 MB) to apply them, which makes
// the forward one kernel and the backward one kernel, with no cross-workgroup reduction at all.  A thread owns 8
// consecutive channels (always inside one group: C / groups is a multiple of 8) of the pixels it visits, so its
// scale / shift, mean / rstd are loop constants.
//
//   forward : xhat = (y - mean_g) * rstd_g ; a = relu(xhat * gamma_c + beta_c) ; dropout ; (max over 2x2)
//   backward: the forward is recomputed from y (no mask, no argmax stored);  with gg the gradient arriving at a,
//             m = HW * C/groups:   dgamma_c += sum gg * xhat,  dbeta_c += sum gg   (per image, folded over images after)
//             dy = rstd_g * (gamma_c * gg - mean_m(gamma * gg) - xhat * mean_m(gamma * gg * xhat))
#include "common.h"
#include "act.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

constexpr int GN_T = 512;          // threads per workgroup
constexpr int GN_MAXC = 1024;
constexpr int GN_MAXG = 32;

struct GnGeom {
    int n_img, H, W, C, G;
    float eps;
};

// per-group mean / rstd of image blockIdx.x from one pass over y; result in st[g][0..1] (LDS) for every thread
__device__ __forceinline__ void gn_image_stats(const bf16_t* __restrict__ yimg, const GnGeom& g, int cshift, int c0,
                                               float* red /*[2][GN_T]*/, float (*st)[2]) {
    const int hw = g.H * g.W;
    float s = 0.f, q = 0.f;
    for (int p = threadIdx.x >> cshift; p < hw; p += GN_T >> cshift) {
        float v[8];
        load8(yimg + (long)p * g.C + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) { s += v[i]; q += v[i] * v[i]; }
    }
    // fold inside the wave first (a fixed shuffle tree): lanes that share a channel group sit cgs apart, and the
    // cpg channel groups of one GroupNorm group are neighbours
    // (with more than 64 channel groups -- C > 512 -- a wave holds only some of the GroupNorm groups: the slots of the
    // others stay zero)
    const int cgs = g.C >> 3, cpg = cgs / g.G;
    for (int o = 32; o >= cgs; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    for (int o = cpg >> 1; o >= 1; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cg = threadIdx.x & (cgs - 1);
    if (cgs > 64) {
        red[threadIdx.x] = 0.f;
        red[GN_T + threadIdx.x] = 0.f;
        __syncthreads();
    }
    if (lane < cgs && (cg & (cpg - 1)) == 0) {
        red[wave * g.G + cg / cpg] = s;
        red[GN_T + wave * g.G + cg / cpg] = q;
    }
    __syncthreads();
    if (threadIdx.x < g.G) {
        double S = 0.0, Q = 0.0;
        for (int w = 0; w < GN_T / 64; ++w) { S += (double)red[w * g.G + threadIdx.x]; Q += (double)red[GN_T + w * g.G + threadIdx.x]; }
        const double m = (double)hw * (g.C / g.G);
        const double mean = S / m;
        double var = Q / m - mean * mean;
        if (var < 0.0) var = 0.0;
        st[threadIdx.x][0] = (float)mean;
        st[threadIdx.x][1] = (float)(1.0 / sqrt(var + (double)g.eps));
    }
    __syncthreads();
}

template <bool POOL>
__global__ __launch_bounds__(GN_T) void gn_act_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, bf16_t* __restrict__ out,
                                                          float* __restrict__ mean_rstd, GnGeom g, DropCfg d) {
    __shared__ float red[2 * GN_T];
    __shared__ float st[GN_MAXG][2];
    const int cgs = g.C >> 3, cshift = __ffs(cgs) - 1;
    const int c0 = (threadIdx.x & (cgs - 1)) * 8;
    const int hw = g.H * g.W;
    const long img = blockIdx.x;
    const bf16_t* yimg = y + img * hw * g.C;
    gn_image_stats(yimg, g, cshift, c0, red, st);
    if (threadIdx.x < g.G && mean_rstd) {
        mean_rstd[(img * g.G + threadIdx.x) * 2] = st[threadIdx.x][0];
        mean_rstd[(img * g.G + threadIdx.x) * 2 + 1] = st[threadIdx.x][1];
    }
    const int grp = c0 / (g.C / g.G);
    const float mean = st[grp][0], rstd = st[grp][1];
    float sc[8], sh[8];
    load8f(gamma + c0, sc);
    load8f(beta + c0, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] *= rstd; sh[i] -= mean * sc[i]; }
    if (!POOL) {
        for (int p = threadIdx.x >> cshift; p < hw; p += GN_T >> cshift) {
            const long idx = (img * hw + p) * g.C + c0;
            float v[8], o[8];
            bool ps[8], kp[8];
            load8(y + idx, v);
            act8(v, sc, sh, d, idx, o, ps, kp);
            store8(out + idx, o);
        }
    } else {
        const int wo_n = g.W >> 1, npo = (g.H >> 1) * wo_n;
        for (int po = threadIdx.x >> cshift; po < npo; po += GN_T >> cshift) {
            const int ho = po / wo_n, wo = po - ho * wo_n;
            float o[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long idx = ((img * g.H + 2 * ho + (k >> 1)) * g.W + 2 * wo + (k & 1)) * g.C + c0;
                float v[8], a[8];
                bool ps[8], kp[8];
                load8(y + idx, v);
                act8(v, sc, sh, d, idx, a, ps, kp);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (k == 0 || a[i] > o[i]) ? a[i] : o[i];
            }
            store8(out + (img * npo + po) * g.C + c0, o);
        }
    }
}

// gradient arriving at the activation of the 8 channels of one (pooling window of) pixel(s), and xhat there.
// !POOL: k = 0 only.  POOL: the first maximum in scan order takes the window's gradient, as in MaxPool2d.
template <bool POOL>
__device__ __forceinline__ void gn_window(const bf16_t* __restrict__ y, const bf16_t* __restrict__ gin, long gidx,
                                          const long idx[4], const float sc[8], const float sh[8], float mean, float rstd,
                                          const DropCfg& d, float gg[4][8], float xh[4][8]) {
    constexpr int K = POOL ? 4 : 1;
    float gi[8], a[4][8];
    bool ps[4][8], kp[4][8];
    load8(gin + gidx, gi);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float v[8];
        load8(y + idx[k], v);
        act8(v, sc, sh, d, idx[k], a[k], ps[k], kp[k]);
#pragma unroll
        for (int i = 0; i < 8; ++i) xh[k][i] = (v[i] - mean) * rstd;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int arg = 0;
        if (POOL) {
            float best = a[0][i];
#pragma unroll
            for (int k = 1; k < K; ++k)
                if (a[k][i] > best) { best = a[k][i]; arg = k; }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float t = (k == arg && ps[k][i]) ? gi[i] : 0.f;
            if (d.enabled) t = kp[k][i] ? t * d.inv_keep : 0.f;
            gg[k][i] = t;
        }
    }
}

template <bool POOL>
__global__ __launch_bounds__(GN_T) void gn_act_bwd_kernel(const bf16_t* __restrict__ gin, const bf16_t* __restrict__ y,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean_rstd,
                                                          float* __restrict__ chan_partial, bf16_t* __restrict__ dy,
                                                          GnGeom g, DropCfg d) {
    __shared__ float red[2][GN_T][8];
    __shared__ float chs[2][GN_MAXC];
    __shared__ float coef[GN_MAXG][2];
    constexpr int K = POOL ? 4 : 1;
    const int cgs = g.C >> 3, cshift = __ffs(cgs) - 1;
    const int c0 = (threadIdx.x & (cgs - 1)) * 8;
    const int hw = g.H * g.W;
    const long img = blockIdx.x;
    const int cpg_ch = g.C / g.G;
    const int grp = c0 / cpg_ch;
    const float mean = mean_rstd[(img * g.G + grp) * 2], rstd = mean_rstd[(img * g.G + grp) * 2 + 1];
    float gm[8], sc[8], sh[8];
    load8f(gamma + c0, gm);
    load8f(beta + c0, sh);
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = gm[i] * rstd; sh[i] -= mean * sc[i]; }
    const int wo_n = g.W >> 1;
    const int npos = POOL ? (g.H >> 1) * wo_n : hw;

    auto locate = [&](int pos, long idx[4], long& gidx) {
        if (!POOL) {
            idx[0] = (img * hw + pos) * g.C + c0;
            gidx = idx[0];
        } else {
            const int ho = pos / wo_n, wo = pos - ho * wo_n;
#pragma unroll
            for (int k = 0; k < 4; ++k) idx[k] = ((img * g.H + 2 * ho + (k >> 1)) * g.W + 2 * wo + (k & 1)) * g.C + c0;
            gidx = (img * npos + pos) * g.C + c0;
        }
    };

    // pass A: per-channel sums of this image
    float a_gx[8], a_g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a_gx[i] = a_g[i] = 0.f;
    for (int pos = threadIdx.x >> cshift; pos < npos; pos += GN_T >> cshift) {
        long idx[4], gidx;
        locate(pos, idx, gidx);
        float gg[4][8], xh[4][8];
        gn_window<POOL>(y, gin, gidx, idx, sc, sh, mean, rstd, d, gg, xh);
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) { a_gx[i] += gg[k][i] * xh[k][i]; a_g[i] += gg[k][i]; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = a_gx[i]; red[1][threadIdx.x][i] = a_g[i]; }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * g.C; j += GN_T) {
        const int row = j / g.C, c = j - row * g.C;
        float s = 0.f;
        for (int t = c >> 3; t < GN_T; t += cgs) s += red[row][t][c & 7];
        chs[row][c] = s;
        chan_partial[(img * 3 + row) * g.C + c] = s;     // [d gamma | d beta] of this image
    }
    __syncthreads();
    if (threadIdx.x < g.G) {
        double A = 0.0, B = 0.0;
        for (int c = threadIdx.x * cpg_ch; c < (threadIdx.x + 1) * cpg_ch; ++c) {
            A += (double)gamma[c] * (double)chs[0][c];
            B += (double)gamma[c] * (double)chs[1][c];
        }
        const double m = (double)hw * cpg_ch;
        coef[threadIdx.x][0] = (float)(A / m);
        coef[threadIdx.x][1] = (float)(B / m);
    }
    __syncthreads();
    const float cA = coef[grp][0], cB = coef[grp][1];

    // pass B: dy, and its per-channel sum (the bias gradient of the conv that produced y)
    float a_dy[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a_dy[i] = 0.f;
    for (int pos = threadIdx.x >> cshift; pos < npos; pos += GN_T >> cshift) {
        long idx[4], gidx;
        locate(pos, idx, gidx);
        float gg[4][8], xh[4][8];
        gn_window<POOL>(y, gin, gidx, idx, sc, sh, mean, rstd, d, gg, xh);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                o[i] = rstd * (gm[i] * gg[k][i] - cB - xh[k][i] * cA);
                a_dy[i] += o[i];
            }
            store8(dy + idx[k], o);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[0][threadIdx.x][i] = a_dy[i];      // (every read of red above is behind two barriers)
    __syncthreads();
    for (int c = threadIdx.x; c < g.C; c += GN_T) {
        float s = 0.f;
        for (int t = c >> 3; t < GN_T; t += cgs) s += red[0][t][c & 7];
        chan_partial[(img * 3 + 2) * g.C + c] = s;
    }
}

static int gn_check(const char* who, int n_img, int H, int W, int C, int G, int pool) {
    if (int e = check_geom(who, n_img, H, W, C, pool)) return e;
    VAE_CHECK(G >= 1 && G <= GN_MAXG && C % (8 * G) == 0, "%s: C=%d must be a multiple of 8 * groups (groups=%d, at most %d)",
              who, C, G, GN_MAXG);
    VAE_CHECK(C <= GN_MAXC && GN_T % (C / 8) == 0 && C / 8 / G <= 64, "%s: C=%d not supported (at most %d)", who, C, GN_MAXC);
    return 0;
}

}  // namespace vae

using namespace vae;

extern "C" int vae_gn_act_fwd(const void* y, const float* gamma, const float* beta, void* out, float* mean_rstd, int n_img,
                              int H, int W, int C, int groups, float eps, int pool, double p_drop, const void* keep_mask,
                              unsigned long long seed, unsigned step, unsigned stream_id, unsigned rank, void* stream) {
    VAE_CHECK(y && gamma && beta && out, "vae_gn_act_fwd: null pointer");
    if (int e = gn_check("vae_gn_act_fwd", n_img, H, W, C, groups, pool)) return e;
    VAE_CHECK(p_drop >= 0.0 && p_drop < 1.0, "vae_gn_act_fwd: p_drop=%f out of [0,1)", p_drop);
    const GnGeom g = {n_img, H, W, C, groups, eps};
    const DropCfg d = make_drop(p_drop, keep_mask, seed, step, stream_id, rank);
    hipStream_t s = (hipStream_t)stream;
    const double elems = (double)n_img * H * W * C;
    ProfScope prof(pool ? "gn_act_fwd_kernel<pool>" : "gn_act_fwd_kernel", 0.0, 2.0 * elems * (pool ? 1.25 : 2.0), s);
    if (pool) hipLaunchKernelGGL(gn_act_fwd_kernel<true>, dim3(n_img), dim3(GN_T), 0, s, (const bf16_t*)y, gamma, beta, (bf16_t*)out, mean_rstd, g, d);
    else hipLaunchKernelGGL(gn_act_fwd_kernel<false>, dim3(n_img), dim3(GN_T), 0, s, (const bf16_t*)y, gamma, beta, (bf16_t*)out, mean_rstd, g, d);
    VAE_LAUNCH_CHECK("gn_act_fwd_kernel");
    return 0;
}

extern "C" int vae_gn_act_bwd(const void* gin, const void* y, const float* gamma, const float* beta, const float* mean_rstd,
                              float* chan_partial, float* sums, void* dy, int n_img, int H, int W, int C, int groups,
                              int pool, double p_drop, const void* keep_mask, unsigned long long seed, unsigned step,
                              unsigned stream_id, unsigned rank, void* stream) {
    VAE_CHECK(gin && y && gamma && beta && mean_rstd && chan_partial && sums && dy, "vae_gn_act_bwd: null pointer");
    if (int e = gn_check("vae_gn_act_bwd", n_img, H, W, C, groups, pool)) return e;
    VAE_CHECK(p_drop >= 0.0 && p_drop < 1.0, "vae_gn_act_bwd: p_drop=%f out of [0,1)", p_drop);
    const GnGeom g = {n_img, H, W, C, groups, 0.f};
    const DropCfg d = make_drop(p_drop, keep_mask, seed, step, stream_id, rank);
    hipStream_t s = (hipStream_t)stream;
    const double elems = (double)n_img * H * W * C;
    {
        ProfScope prof(pool ? "gn_act_bwd_kernel<pool>" : "gn_act_bwd_kernel", 0.0, 2.0 * elems * (2.0 + (pool ? 0.25 : 1.0)), s);
        if (pool) hipLaunchKernelGGL(gn_act_bwd_kernel<true>, dim3(n_img), dim3(GN_T), 0, s, (const bf16_t*)gin, (const bf16_t*)y, gamma, beta, mean_rstd, chan_partial, (bf16_t*)dy, g, d);
        else hipLaunchKernelGGL(gn_act_bwd_kernel<false>, dim3(n_img), dim3(GN_T), 0, s, (const bf16_t*)gin, (const bf16_t*)y, gamma, beta, mean_rstd, chan_partial, (bf16_t*)dy, g, d);
        VAE_LAUNCH_CHECK("gn_act_bwd_kernel");
    }
    // [d gamma | d beta | d bias] = the per-image rows folded over the images (fixed order)
    return vae_sum_partials(chan_partial, n_img, 3 * C, sums, 1.0f, stream);
}
