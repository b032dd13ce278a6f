// HBM-bound glue of the conv VAE: BatchNorm (train/eval) + ReLU + Dropout + MaxPool2d(2)
// forward and backward, the 1x1 head + sigmoid, tanh backward, layout conversion, and the
// small partial-sum reducers.  Activations are NHWC bf16; one thread owns 8 consecutive
// channels (one 16-byte access) of a pixel, or of a 2x2 pooling window.
//
//   forward : a = relu(y*scale + shift); d = keep ? a/(1-p) : 0; out = d or max over the window
//   backward: the forward is RECOMPUTED from the saved pre-norm tensor y (no mask, no argmax and
//             no full-resolution activation is stored): pass 1 reduces sum(g) and sum(g*xhat) per
//             channel, pass 2 writes dy = scale * (g - mean(g) - xhat * mean(g*xhat)).
#include "common.h"
#include "act.h"
#include <type_traits>
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

// ---------------------------------------------------------------------------------------
// batch statistics -> scale / shift / mean / rstd (+ running stats), one pass over the
// per-workgroup partials left by the GEMM epilogue.  grid = ceil(C/8), block = 1024 = 8 channels x 128 row
// groups: the launch is latency-bound, so the few hundred KB of partials are spread over as many CUs as there are
// 32-byte column slices.
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float* __restrict__ psum, const float* __restrict__ psq, int rows, int C, int fold, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, float eps, int use_running, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double red[2][128][9];
    const int cl = threadIdx.x & 7, rg = threadIdx.x >> 3;      // 8 channels x 128 row groups
    const int c = blockIdx.x * 8 + cl;
    double s = 0.0, q = 0.0;
    if (!use_running && c < C) {
        const int ncols = fold * C;
        for (int r = rg; r < rows; r += 128)
            for (int t = 0; t < fold; ++t) {
                s += (double)psum[(long)r * ncols + t * C + c];
                q += (double)psq[(long)r * ncols + t * C + c];
            }
    }
    red[0][rg][cl] = s;
    red[1][rg][cl] = q;
    __syncthreads();
    if (rg < 8) {          // fixed-order tree: 128 row groups -> 8 -> 1
        s = 0.0; q = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s += red[0][rg + 8 * i][cl]; q += red[1][rg + 8 * i][cl]; }
    }
    __syncthreads();
    if (rg < 8) { red[0][rg][cl] = s; red[1][rg][cl] = q; }
    __syncthreads();
    if (rg == 0 && c < C) {
        float mean, var;
        if (use_running) {
            mean = running_mean[c];
            var = running_var[c];
        } else {
#pragma unroll
            for (int i = 1; i < 8; ++i) { s += red[0][i][cl]; q += red[1][i][cl]; }
            const double m = s / count;
            double v = q / count - m * m;
            if (v < 0.0) v = 0.0;
            mean = (float)m;
            var = (float)v;
            if (running_mean) {
                const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
        const float rstd = 1.0f / sqrtf(var + eps);
        const float sc = gamma[c] * rstd;
        scale[c] = sc;
        shift[c] = beta[c] - mean * sc;
        mean_out[c] = mean;
        rstd_out[c] = rstd;
    }
}

// Stage 1 of the statistics fold when there are many partial rows (one CU cannot stream MBs of
// partials fast enough): grid = (ceil(ncols/64), 32) workgroups each sum rows s, s+32, s+64, ...
// of BOTH arrays into scratch row `rows + s` (the 32 rows past the partials).
__global__ __launch_bounds__(256) void stats_stage1_kernel(float* __restrict__ psum, float* __restrict__ psq, int rows,
                                                           int ncols) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f, q = 0.f;
    if (c < ncols) {
        int r = blockIdx.y + 32 * rg;
        for (; r + 384 < rows; r += 512) {          // four rows of both arrays in flight
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = psum[(long)(r + 128 * i) * ncols + c]; b[i] = psq[(long)(r + 128 * i) * ncols + c]; }
            s += (a[0] + a[1]) + (a[2] + a[3]);
            q += (b[0] + b[1]) + (b[2] + b[3]);
        }
        for (; r < rows; r += 128) {
            s += psum[(long)r * ncols + c];
            q += psq[(long)r * ncols + c];
        }
    }
    red[0][rg][cl] = s;
    red[1][rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < ncols) {
        psum[(long)(rows + blockIdx.y) * ncols + c] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        psq[(long)(rows + blockIdx.y) * ncols + c] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

// out[j] = sum_r partial[r][j]; grid = ceil(ncols/32), block = 1024 = 32 columns x 32 row groups (a launch of
// this kernel is latency-bound: narrow column slices put more CUs on the few hundred KB of partials)
__global__ __launch_bounds__(1024) void sum_partials_kernel(const float* __restrict__ partial, int rows, int ncols,
                                                            int ld, float* __restrict__ out, float mul) {
    __shared__ double red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < ncols) {
        int r = rg;
        for (; r + 224 < rows; r += 256) {         // eight loads in flight per lane: the launch is a chain of round trips
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = partial[(long)(r + 32 * i) * ld + c];
            s0 += (double)v[0] + (double)v[4];
            s1 += (double)v[1] + (double)v[5];
            s2 += (double)v[2] + (double)v[6];
            s3 += (double)v[3] + (double)v[7];
        }
        for (; r + 32 < rows; r += 64) {
            s0 += (double)partial[(long)r * ld + c];
            s1 += (double)partial[(long)(r + 32) * ld + c];
        }
        if (r < rows) s0 += (double)partial[(long)r * ld + c];
    }
    red[rg][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && c < ncols) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += red[i][cl];
        out[c] = (float)(s * (double)mul);
    }
}

// ---------------------------------------------------------------------------------------
int g_pool_ymax_mode = 0;
int g_head_sums_mode = 0;
// bn_conv1_bwd.hip
long head_sums_scratch_floats(long npix);
int head_sums_mfma(const float* drecon, const float* recon, const bf16_t* y, const float* scale, const float* shift,
                   const float* mean, const float* rstd, const float* w, float* scratch, float* sums, float* sums_head,
                   long npix, int hw, int CO, float* dl_out, hipStream_t s);

static int pow2_shift(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

struct ActGeom {
    int n_img, H, W, C;   // full-resolution geometry of y
    int pool;
    int wsh2, hsh2;       // log2(W/2), log2(H/2) when both are powers of two, else -1
};

// pooled position -> (img, ho, wo); shifts on power-of-two grids, 32-bit division otherwise
__device__ __forceinline__ void split_pooled(const ActGeom& g, unsigned pos, int& wo, int& ho, long& img) {
    if (g.wsh2 >= 0 && g.hsh2 >= 0) {
        wo = pos & ((g.W >> 1) - 1);
        ho = (pos >> g.wsh2) & ((g.H >> 1) - 1);
        img = pos >> (g.wsh2 + g.hsh2);
    } else {
        wo = pos % (unsigned)(g.W >> 1);
        const unsigned t = pos / (unsigned)(g.W >> 1);
        ho = t % (unsigned)(g.H >> 1);
        img = t / (unsigned)(g.H >> 1);
    }
}

// POOL with `route`: one byte per pooled element records what the backward pass needs to know about the window --
// bits 0-1 the position (scan order, first maximum wins) of the maximum, bit 2 set when that maximum is positive
// (the element was kept by dropout and passed the ReLU, so the gradient flows) -- so the backward kernels neither
// regenerate the dropout stream nor redo the activation of the four window elements.
template <bool POOL>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, bf16_t* __restrict__ out,
                                                         uint8_t* __restrict__ route, bf16_t* __restrict__ ymax, ActGeom g,
                                                         DropCfg d) {
    const int cgs = g.C >> 3;
    const long npos = POOL ? (long)g.n_img * (g.H >> 1) * (g.W >> 1) : (long)g.n_img * g.H * g.W;
    // cgs is a power of two dividing 256: a thread keeps its channel group, positions advance by a fixed stride
    const int cshift = __ffs(cgs) - 1;
    const int c0 = (threadIdx.x & (cgs - 1)) * 8;
    float sc[8], sh[8];
    load8f(scale + c0, sc);
    load8f(shift + c0, sh);
    const unsigned pstride = (gridDim.x * 256u) >> cshift;
    for (unsigned pos = (blockIdx.x * 256u + threadIdx.x) >> cshift; pos < npos; pos += pstride) {
        float o[8];
        bool ps[8], kp[8];
        if (!POOL) {
            float v[8];
            load8s(y + (long)pos * g.C + c0, v);
            act8(v, sc, sh, d, (long)pos * g.C + c0, o, ps, kp);
        } else {
            int wo, ho;
            long img;
            split_pooled(g, pos, wo, ho, img);
            // The scan keeps, per channel, the running maximum and ONE word beside it: the bits of the conv output there
            // (bf16, so the low half of the word is free) with the window position in bits 0-1.  Keep flags become a
            // multiplier (1 / keep or 0) instead of a lane mask: no mask registers, one select per element.
            uint32_t ya[8];
            auto scan = [&](auto drop) {
                constexpr bool DROP = decltype(drop)::value;
                // the window's four loads go out together, ahead of the generator's arithmetic
                const long e00 = ((img * g.H + 2 * ho) * g.W + 2 * wo) * g.C + c0;
                u32x4s uu[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    uu[k] = __builtin_nontemporal_load((const u32x4s*)(y + e00 + ((k >> 1) * g.W + (k & 1)) * g.C));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long e0 = e00 + ((k >> 1) * g.W + (k & 1)) * g.C;
                    const u32x4s u = uu[k];
                    float km[8];
                    if (DROP) {
                        if (d.mask) {
                            const uint2 m = *(const uint2*)(d.mask + e0);
#pragma unroll
                            for (int i = 0; i < 8; ++i) km[i] = (((i < 4 ? m.x : m.y) >> (8 * (i & 3))) & 0xFFu) ? d.inv_keep : 0.f;
                        } else {
                            uint32_t w[4];
                            philox_block(d.key, (uint32_t)(e0 >> 3), w);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                km[2 * i] = (w[i] & 0xFFFFu) >= d.thr ? d.inv_keep : 0.f;
                                km[2 * i + 1] = (w[i] >> 16) >= d.thr ? d.inv_keep : 0.f;
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint32_t vb = (i & 1) ? (u[i >> 1] & 0xFFFF0000u) : (u[i >> 1] << 16);
                        const float a = __uint_as_float(vb) * sc[i] + sh[i];
                        float r = a > 0.f ? a : 0.f;
                        if (DROP) r *= km[i];
                        if (k == 0) { o[i] = r; ya[i] = vb; }
                        else {
                            const bool c = r > o[i];        // strict >: the first maximum wins, as in MaxPool2d
                            o[i] = c ? r : o[i];
                            ya[i] = c ? (vb | (uint32_t)k) : ya[i];
                        }
                    }
                }
            };
            if (d.enabled) scan(std::true_type{});
            else scan(std::false_type{});
            if (ymax) {         // exact: the values are bf16 already
                const u32x4s q = {(ya[0] >> 16) | (ya[1] & 0xFFFF0000u), (ya[2] >> 16) | (ya[3] & 0xFFFF0000u),
                                  (ya[4] >> 16) | (ya[5] & 0xFFFF0000u), (ya[6] >> 16) | (ya[7] & 0xFFFF0000u)};
                __builtin_nontemporal_store(q, (u32x4s*)(ymax + (long)pos * g.C + c0));
            }
            if (route) {
                uint32_t r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) r[i] = (ya[i] & 3u) | (o[i] > 0.f ? 4u : 0u);
                *(uint2*)(route + (long)pos * g.C + c0) = make_uint2(r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24),
                                                                      r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24));
            }
        }
        store8s(out + (long)pos * g.C + c0, o);
    }
}

// PASS 0: reduce sum(g*xhat), sum(g) per channel into partial[block][2][C];  PASS 1: write dy.
// ROUTED (POOL only): the forward pass left one route byte per pooled element (see bn_act_fwd_kernel).
template <bool POOL, int PASS, bool ROUTED>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(const bf16_t* __restrict__ gin, const bf16_t* __restrict__ y,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ sums, float inv_count,
                                                         float* __restrict__ partial, bf16_t* __restrict__ dy,
                                                         const uint8_t* __restrict__ route, const bf16_t* __restrict__ ymax,
                                                         ActGeom g, DropCfg d) {
    __shared__ float red[2][256][8];
    const int cgs = g.C >> 3;
    const long npos = POOL ? (long)g.n_img * (g.H >> 1) * (g.W >> 1) : (long)g.n_img * g.H * g.W;
    const int cshift = __ffs(cgs) - 1;      // cgs is a power of two dividing 256
    const int cg = threadIdx.x & (cgs - 1); // fixed per thread: block and grid strides are multiples of cgs
    const int c0 = cg * 8;
    float sc[8], sh[8], mu[8], rs[8], c1[8], c2[8];
    load8f(scale + c0, sc);
    load8f(shift + c0, sh);
    load8f(mean + c0, mu);
    load8f(rstd + c0, rs);
    if (PASS == 1) {
        load8f(sums + c0, c2);            // sum(g*xhat)
        load8f(sums + g.C + c0, c1);      // sum(g)
#pragma unroll
        for (int i = 0; i < 8; ++i) { c1[i] *= inv_count; c2[i] *= inv_count; }
    }
    float a_gx[8], a_g[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a_gx[i] = a_g[i] = 0.f;

    const unsigned pstride = (gridDim.x * 256u) >> cshift;
    for (unsigned up = (blockIdx.x * 256u + threadIdx.x) >> cshift; up < npos; up += pstride) {
        const long pos = up;
        float gi[8];
        if (PASS == 1) load8s(gin + pos * g.C + c0, gi);      // second and last read
        else load8(gin + pos * g.C + c0, gi);
        if (!POOL) {
            float v[8], a[8];
            bool ps[8], kp[8];
            if (PASS == 1) load8s(y + pos * g.C + c0, v);
            else load8(y + pos * g.C + c0, v);
            act8(v, sc, sh, d, pos * g.C + c0, a, ps, kp);
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float gg = ps[i] ? gi[i] : 0.f;
                if (d.enabled) gg = kp[i] ? gg * d.inv_keep : 0.f;
                const float xh = (v[i] - mu[i]) * rs[i];
                if (PASS == 0) { a_gx[i] += gg * xh; a_g[i] += gg; }
                else o[i] = sc[i] * (gg - c1[i] - xh * c2[i]);
            }
            if (PASS == 1) store8s(dy + pos * g.C + c0, o);
        } else if (ROUTED) {
            int wo, ho;
            long img;
            split_pooled(g, up, wo, ho, img);
            const uint2 rr = *(const uint2*)(route + pos * g.C + c0);
            if (PASS == 0 && ymax) {
                // the forward pass left the conv output at each window's maximum: a quarter of the reads, the same values
                float vs[8];
                load8(ymax + pos * g.C + c0, vs);
                const float ik0 = d.enabled ? d.inv_keep : 1.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t r = ((i < 4 ? rr.x : rr.y) >> (8 * (i & 3))) & 0xFFu;
                    const float gg = (r & 4u) ? gi[i] * ik0 : 0.f;
                    a_gx[i] += gg * ((vs[i] - mu[i]) * rs[i]);
                    a_g[i] += gg;
                }
                continue;
            }
            float v[4][8];
            long pix[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pix[k] = (img * g.H + 2 * ho + (k >> 1)) * g.W + 2 * wo + (k & 1);
                if (PASS == 1) load8s(y + pix[k] * g.C + c0, v[k]);
                else load8(y + pix[k] * g.C + c0, v[k]);
            }
            const float ik = d.enabled ? d.inv_keep : 1.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t r = ((i < 4 ? rr.x : rr.y) >> (8 * (i & 3))) & 0xFFu;
                const uint32_t arg = r & 3u;
                const float gg = (r & 4u) ? gi[i] * ik : 0.f;          // gradient arriving at the window's maximum
                if (PASS == 0) {
                    const float vs = arg == 0 ? v[0][i] : arg == 1 ? v[1][i] : arg == 2 ? v[2][i] : v[3][i];
                    a_gx[i] += gg * ((vs - mu[i]) * rs[i]);
                    a_g[i] += gg;
                } else {
                    // dx = sc * (g - c1 - xhat * c2): the g term only at the maximum
                    const float a0 = -sc[i] * c1[i], b0 = -sc[i] * c2[i] * rs[i], sg = sc[i] * gg;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k][i] = (v[k][i] - mu[i]) * b0 + a0 + (arg == (uint32_t)k ? sg : 0.f);
                }
            }
            if (PASS == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) store8s(dy + pix[k] * g.C + c0, v[k]);
            }
        } else {
            int wo, ho;
            long img;
            split_pooled(g, up, wo, ho, img);
            float v[4][8], a[4][8];
            bool ps[4][8], kp[4][8];
            long pix[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pix[k] = (img * g.H + 2 * ho + (k >> 1)) * g.W + 2 * wo + (k & 1);
                load8(y + pix[k] * g.C + c0, v[k]);
                act8(v[k], sc, sh, d, pix[k] * g.C + c0, a[k], ps[k], kp[k]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // first maximum in scan order wins (strict >), as in MaxPool2d
                int arg = 0;
                float best = a[0][i];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (a[k][i] > best) { best = a[k][i]; arg = k; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float gg = (k == arg && ps[k][i]) ? gi[i] : 0.f;
                    if (d.enabled) gg = kp[k][i] ? gg * d.inv_keep : 0.f;
                    const float xh = (v[k][i] - mu[i]) * rs[i];
                    if (PASS == 0) { a_gx[i] += gg * xh; a_g[i] += gg; }
                    else a[k][i] = sc[i] * (gg - c1[i] - xh * c2[i]);
                }
            }
            if (PASS == 1) {
#pragma unroll
                for (int k = 0; k < 4; ++k) store8s(dy + pix[k] * g.C + c0, a[k]);
            }
        }
    }
    if (PASS == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = a_gx[i]; red[1][threadIdx.x][i] = a_g[i]; }
        __syncthreads();
        // threads 0 .. 2*C-1 each fold one (row, channel) column over the pixel lanes
        for (int j = threadIdx.x; j < 2 * g.C; j += blockDim.x) {
            const int row = j / g.C, c = j % g.C;
            float s = 0.f;
            for (int t = (c >> 3); t < 256; t += cgs) s += red[row][t][c & 7];
            partial[(long)blockIdx.x * 2 * g.C + j] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------
// 1x1 head + sigmoid fused with the last BatchNorm + ReLU.  8 lanes share one pixel.
//   a = relu(y*scale + shift) ; logit[o] = sum_c a[c] w[o][c] + b[o] ; recon = sigmoid(logit)
template <int CO>
__global__ __launch_bounds__(256) void head_fwd_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ recon,
                                                       long npix, int hw, int hwshift) {
    const int sub = threadIdx.x & 7, c0 = sub * 8;
    float sc[8], sh[8], wv[CO][8];
    load8f(scale + c0, sc);
    load8f(shift + c0, sh);
#pragma unroll
    for (int o = 0; o < CO; ++o) load8f(w + o * 64 + c0, wv[o]);
    const long stride = (long)gridDim.x * (blockDim.x >> 3);
    const long iters = (npix + stride - 1) / stride;
    // HU pixels per lane and round: their loads are all in flight before any is used
    constexpr int HU = 4;
    for (long it = 0; it < iters; it += HU) {
        long pix[HU];
        bool ok[HU];
        uint4 raw[HU];
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            pix[u] = (it + u) * stride + (long)blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3);
            ok[u] = pix[u] < npix;
            raw[u] = make_uint4(0u, 0u, 0u, 0u);
            if (ok[u]) raw[u] = *(const uint4*)(y + pix[u] * 64 + c0);      // plain: the backward pass re-reads y a few kernels later, from cache
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            float acc[CO];
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[o] = 0.f;
            const uint32_t wd[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float vi = __uint_as_float((i & 1) ? (wd[i >> 1] & 0xFFFF0000u) : (wd[i >> 1] << 16));
                const float a = fmaxf(vi * sc[i] + sh[i], 0.f);
#pragma unroll
                for (int o = 0; o < CO; ++o) acc[o] += a * wv[o][i];
            }
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                acc[o] += __shfl_xor(acc[o], 1, 64);
                acc[o] += __shfl_xor(acc[o], 2, 64);
                acc[o] += __shfl_xor(acc[o], 4, 64);
            }
            if (ok[u] && sub < CO) {
                float l = 0.f;
#pragma unroll
                for (int o = 0; o < CO; ++o) if (sub == o) l = acc[o] + b[o];
                // npix < 2^31: 32-bit index math, shifts when the image is a power of two
                const unsigned up = (unsigned)pix[u];
                const unsigned img = hwshift >= 0 ? up >> hwshift : up / (unsigned)hw, p = up - img * (unsigned)hw;
                recon[((long)img * CO + sub) * hw + p] = 1.f / (1.f + expf(-l));
            }
        }
    }
}

// PASS 0: partial[block] = [sum(g*xhat)(64) | sum(g)(64) | dW[CO][64] | db[CO] (padded to 64)]
// PASS 1: dy = scale * (g - mean(g) - xhat*mean(g*xhat))
template <int CO, int PASS>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ drecon, const float* __restrict__ recon,
                                                       const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ w,
                                                       const float* __restrict__ sums, float inv_count,
                                                       float* __restrict__ partial, bf16_t* __restrict__ dy, long npix,
                                                       int hw, int hwshift) {
    constexpr int ROWS = 3 + CO;
    // per-wave partial sums only (the eight pixel lanes of a wave are folded by shuffles first): 6 KB instead of the
    // 50 KB a [row][pixel lane][channel] array takes, which held a CU to three workgroups
    __shared__ float red[ROWS][4][64];
    const int sub = threadIdx.x & 7, c0 = sub * 8, pl = threadIdx.x >> 3;
    float sc[8], sh[8], mu[8], rs[8], wv[CO][8], c1[8], c2[8];
    load8f(scale + c0, sc);
    load8f(shift + c0, sh);
    load8f(mean + c0, mu);
    load8f(rstd + c0, rs);
#pragma unroll
    for (int o = 0; o < CO; ++o) load8f(w + o * 64 + c0, wv[o]);
    if (PASS == 1) {
        load8f(sums + c0, c2);
        load8f(sums + 64 + c0, c1);
#pragma unroll
        for (int i = 0; i < 8; ++i) { c1[i] *= inv_count; c2[i] *= inv_count; }
    }
    float a_gx[8], a_g[8], a_w[CO][8], a_b[CO];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a_gx[i] = a_g[i] = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) a_w[o][i] = 0.f;
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) a_b[o] = 0.f;

    // two pixels per trip: both pixels' loads are issued before either is used (the loop is one dependent chain
    // per lane otherwise, and a launch this wide lives on loads in flight)
    const long stride = (long)gridDim.x * 32;
    for (long pix0 = (long)blockIdx.x * 32 + pl; pix0 < npix; pix0 += 2 * stride) {
        float vv[2][8], dd[2][CO];
        bool okk[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long pix = pix0 + u * stride;
            okk[u] = pix < npix;
            if (okk[u]) {
                const unsigned up = (unsigned)pix;      // npix < 2^31
                const unsigned img = hwshift >= 0 ? up >> hwshift : up / (unsigned)hw, p = up - img * (unsigned)hw;
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const long e = ((long)img * CO + o) * hw + p;
                    const float r = recon[e];
                    dd[u][o] = drecon[e] * r * (1.f - r);
                }
                load8(y + pix * 64 + c0, vv[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
        if (!okk[u]) continue;
        const long pix = pix0 + u * stride;
        float v[8], o8[8], dl[CO];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = vv[u][i];
#pragma unroll
        for (int o = 0; o < CO; ++o) dl[o] = dd[u][o];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float a = v[i] * sc[i] + sh[i];
            const bool pos = a > 0.f;
            float da = 0.f;
#pragma unroll
            for (int o = 0; o < CO; ++o) da += dl[o] * wv[o][i];
            const float gg = pos ? da : 0.f;
            const float xh = (v[i] - mu[i]) * rs[i];
            if (PASS == 0) {
                a_gx[i] += gg * xh;
                a_g[i] += gg;
                const float ar = pos ? a : 0.f;
#pragma unroll
                for (int o = 0; o < CO; ++o) a_w[o][i] += dl[o] * ar;
            } else {
                o8[i] = sc[i] * (gg - c1[i] - xh * c2[i]);
            }
        }
        if (PASS == 0 && sub == 0) {
#pragma unroll
            for (int o = 0; o < CO; ++o) a_b[o] += dl[o];
        }
        if (PASS == 1) store8(dy + pix * 64 + c0, o8);
        }
    }
    if (PASS == 0) {
        // lanes l, l ^ 8, l ^ 16, l ^ 32 hold the same channels of different pixels
        auto fold8 = [](float v) {
            v += __shfl_xor(v, 8, 64);
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            return v;
        };
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a_gx[i] = fold8(a_gx[i]);
            a_g[i] = fold8(a_g[i]);
#pragma unroll
            for (int o = 0; o < CO; ++o) a_w[o][i] = fold8(a_w[o][i]);
        }
#pragma unroll
        for (int o = 0; o < CO; ++o) a_b[o] = fold8(a_b[o]);
        const int wv_i = threadIdx.x >> 6;
        if ((threadIdx.x & 63) < 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                red[0][wv_i][c0 + i] = a_gx[i];
                red[1][wv_i][c0 + i] = a_g[i];
#pragma unroll
                for (int o = 0; o < CO; ++o) red[2 + o][wv_i][c0 + i] = a_w[o][i];
            }
            if (sub == 0) {
#pragma unroll
                for (int o = 0; o < CO; ++o) red[2 + CO][wv_i][o] = a_b[o];
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < ROWS * 64; j += blockDim.x) {
            const int row = j >> 6, c = j & 63;
            float s = 0.f;
            if (row < 2 + CO || c < CO)
                for (int t = 0; t < 4; ++t) s += red[row][t][c];
            partial[(long)blockIdx.x * ROWS * 64 + j] = s;
        }
    }
}

// d(pre-tanh) = dx * (1 - h^2); also per-column partial sums (the fc_up bias gradient)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const bf16_t* __restrict__ dx, const bf16_t* __restrict__ h,
                                                       bf16_t* __restrict__ dpre, float* __restrict__ colpartial,
                                                       int rows, int cols) {
    // one thread per 8 columns, grid.y strides over rows
    const int c0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (c0 >= cols) return;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        float a[8], b[8], o[8];
        load8(dx + (long)r * cols + c0, a);
        load8(h + (long)r * cols + c0, b);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o[i] = a[i] * (1.f - b[i] * b[i]);
            acc[i] += o[i];
        }
        store8(dpre + (long)r * cols + c0, o);
    }
    *(float4*)(colpartial + (long)blockIdx.y * cols + c0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *(float4*)(colpartial + (long)blockIdx.y * cols + c0 + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// network input NCHW fp32 (C <= 8) -> NHWC bf16 padded to 8 channels
__global__ __launch_bounds__(256) void nchw_to_nhwc8_kernel(const float* __restrict__ x, bf16_t* __restrict__ out,
                                                            long npix, int C, int hw, long long* __restrict__ counters,
                                                            int ncount) {
    if (counters && blockIdx.x == 0 && (int)threadIdx.x < ncount) counters[threadIdx.x] += 1;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
        const long img = pix / hw, p = pix % hw;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = c < C ? x[(img * C + c) * hw + p] : 0.f;
        store8(out + pix * 8, v);
    }
}

// Workgroups for a streaming kernel over `items` 16-byte lane items: two items per lane.  (Measured on a 3 x 256 MiB
// stream: as many short workgroups as there are runs reach 6.3 TB/s, a grid capped at 2048 workgroups walking the
// same data in a grid-stride loop 5.0-5.3.)
static int elem_grid(long items) {
    long b = (items + 511) / 512;
    if (b < 1) b = 1;
    if (b > (1L << 22)) b = 1L << 22;
    return (int)b;
}

}  // namespace vae

using namespace vae;

extern "C" int vae_bn_finalize(float* stat_sum, float* stat_sq, int rows, int C, int fold, double count,
                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, int use_running, float* scale, float* shift, float* mean,
                               float* rstd, void* stream) {
    VAE_CHECK(gamma && beta && scale && shift && mean && rstd, "vae_bn_finalize: null pointer");
    VAE_CHECK(use_running ? (running_mean && running_var) : (stat_sum && stat_sq && rows > 0 && count > 0),
              "vae_bn_finalize: missing statistics input");
    VAE_CHECK(C > 0 && fold > 0, "vae_bn_finalize: bad C=%d fold=%d", C, fold);
    if (!use_running && (long)rows * fold > 1024) {
        // many partial rows: fold them to 32 scratch rows with the whole chip first
        const int ncols = fold * C;
        hipLaunchKernelGGL(stats_stage1_kernel, dim3((ncols + 63) / 64, VAE_STAT_SCRATCH_ROWS), dim3(256), 0,
                           (hipStream_t)stream, stat_sum, stat_sq, rows, ncols);
        VAE_LAUNCH_CHECK("stats_stage1_kernel");
        stat_sum += (long)rows * ncols;
        stat_sq += (long)rows * ncols;
        rows = VAE_STAT_SCRATCH_ROWS;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(1024), 0, (hipStream_t)stream, stat_sum, stat_sq,
                       rows, C, fold, count, gamma, beta, running_mean, running_var, momentum, eps, use_running, scale,
                       shift, mean, rstd);
    VAE_LAUNCH_CHECK("bn_finalize_kernel");
    return 0;
}

extern "C" int vae_sum_partials(const float* partial, int rows, int ncols, float* out, float mul, void* stream) {
    VAE_CHECK(partial && out && rows > 0 && ncols > 0, "vae_sum_partials: bad arguments");
    hipLaunchKernelGGL(sum_partials_kernel, dim3((ncols + 31) / 32), dim3(1024), 0, (hipStream_t)stream, partial, rows,
                       ncols, ncols, out, mul);
    VAE_LAUNCH_CHECK("sum_partials_kernel");
    return 0;
}

extern "C" int vae_bn_act_fwd(const void* y, const float* scale, const float* shift, void* out, int n_img, int H, int W,
                              int C, int pool, double p_drop, const void* keep_mask, unsigned long long seed,
                              unsigned step, unsigned stream_id, unsigned rank, void* stream) {
    return vae_bn_act_fwd_route(y, scale, shift, out, nullptr, n_img, H, W, C, pool, p_drop, keep_mask, seed, step,
                                stream_id, rank, stream);
}

extern "C" int vae_bn_act_fwd_route(const void* y, const float* scale, const float* shift, void* out, void* route,
                                    int n_img, int H, int W, int C, int pool, double p_drop, const void* keep_mask,
                                    unsigned long long seed, unsigned step, unsigned stream_id, unsigned rank,
                                    void* stream) {
    return vae_bn_act_fwd_route_y(y, scale, shift, out, route, nullptr, n_img, H, W, C, pool, p_drop, keep_mask, seed, step,
                                  stream_id, rank, stream);
}

extern "C" int vae_bn_act_fwd_route_y(const void* y, const float* scale, const float* shift, void* out, void* route,
                                      void* ymax, int n_img, int H, int W, int C, int pool, double p_drop,
                                      const void* keep_mask, unsigned long long seed, unsigned step, unsigned stream_id,
                                      unsigned rank, void* stream) {
    VAE_CHECK(!ymax || route, "vae_bn_act_fwd_route_y: ymax goes with the route bytes");
    VAE_CHECK(((uintptr_t)ymax & 15) == 0, "vae_bn_act_fwd_route_y: ymax must be 16-byte aligned");
    VAE_CHECK(y && scale && shift && out, "vae_bn_act_fwd: null pointer");
    VAE_CHECK(!route || pool, "vae_bn_act_fwd_route: route bytes exist only for the pooled form");
    VAE_CHECK(((uintptr_t)route & 7) == 0, "vae_bn_act_fwd_route: route must be 8-byte aligned");
    if (int e = check_geom("vae_bn_act_fwd", n_img, H, W, C, pool)) return e;
    VAE_CHECK(p_drop >= 0.0 && p_drop < 1.0, "vae_bn_act_fwd: p_drop=%f out of [0,1)", p_drop);
    ActGeom g = {n_img, H, W, C, pool, pow2_shift(W >> 1), pow2_shift(H >> 1)};
    const DropCfg d = make_drop(p_drop, keep_mask, seed, step, stream_id, rank);
    const long items = (long)n_img * H * W / (pool ? 4 : 1) * (C / 8);
    hipStream_t s = (hipStream_t)stream;
    const double elems = (double)n_img * H * W * C;
    ProfScope prof(pool ? "bn_act_fwd_kernel<pool>" : "bn_act_fwd_kernel", 0.0, 2.0 * elems * (pool ? (ymax ? 1.5 : 1.25) : 2.0), s);
    if (pool) hipLaunchKernelGGL(bn_act_fwd_kernel<true>, dim3(elem_grid(items)), dim3(256), 0, s, (const bf16_t*)y, scale, shift, (bf16_t*)out, (uint8_t*)route, (bf16_t*)ymax, g, d);
    else hipLaunchKernelGGL(bn_act_fwd_kernel<false>, dim3(elem_grid(items)), dim3(256), 0, s, (const bf16_t*)y, scale, shift, (bf16_t*)out, (uint8_t*)nullptr, (bf16_t*)nullptr, g, d);
    VAE_LAUNCH_CHECK("bn_act_fwd_kernel");
    return 0;
}

extern "C" int vae_bn_act_bwd_blocks(int n_img, int H, int W, int C, int pool) {
    const long items = (long)n_img * H * W / (pool ? 4 : 1) * (C / 8);
    long b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (int)b;
}

extern "C" int vae_bn_act_bwd(const void* gin, const void* y, const float* scale, const float* shift, const float* mean,
                              const float* rstd, float* partial, float* sums, void* dy, int n_img, int H, int W, int C,
                              int pool, double p_drop, const void* keep_mask, unsigned long long seed, unsigned step,
                              unsigned stream_id, unsigned rank, void* stream) {
    return vae_bn_act_bwd_route(gin, y, scale, shift, mean, rstd, partial, sums, dy, nullptr, n_img, H, W, C, pool, p_drop,
                                keep_mask, seed, step, stream_id, rank, stream);
}

extern "C" int vae_bn_act_bwd_route(const void* gin, const void* y, const float* scale, const float* shift,
                                    const float* mean, const float* rstd, float* partial, float* sums, void* dy,
                                    const void* route, int n_img, int H, int W, int C, int pool, double p_drop,
                                    const void* keep_mask, unsigned long long seed, unsigned step, unsigned stream_id,
                                    unsigned rank, void* stream) {
    return vae_bn_act_bwd_route_y(gin, y, scale, shift, mean, rstd, partial, sums, dy, route, nullptr, n_img, H, W, C, pool,
                                  p_drop, keep_mask, seed, step, stream_id, rank, stream);
}

extern "C" int vae_bn_act_bwd_route_y(const void* gin, const void* y, const float* scale, const float* shift,
                                      const float* mean, const float* rstd, float* partial, float* sums, void* dy,
                                      const void* route, const void* ymax, int n_img, int H, int W, int C, int pool,
                                      double p_drop, const void* keep_mask, unsigned long long seed, unsigned step,
                                      unsigned stream_id, unsigned rank, void* stream) {
    VAE_CHECK(!ymax || route, "vae_bn_act_bwd_route_y: ymax goes with the route bytes");
    VAE_CHECK(((uintptr_t)ymax & 15) == 0, "vae_bn_act_bwd_route_y: ymax must be 16-byte aligned");
    const bf16_t* ym = (const bf16_t*)ymax;
    VAE_CHECK(gin && y && scale && shift && mean && rstd && partial && sums && dy, "vae_bn_act_bwd: null pointer");
    VAE_CHECK(!route || pool, "vae_bn_act_bwd_route: route bytes exist only for the pooled form");
    VAE_CHECK(((uintptr_t)route & 7) == 0, "vae_bn_act_bwd_route: route must be 8-byte aligned");
    const uint8_t* rt = (const uint8_t*)route;
    if (int e = check_geom("vae_bn_act_bwd", n_img, H, W, C, pool)) return e;
    ActGeom g = {n_img, H, W, C, pool, pow2_shift(W >> 1), pow2_shift(H >> 1)};
    const DropCfg d = make_drop(p_drop, keep_mask, seed, step, stream_id, rank);
    const int nb = vae_bn_act_bwd_blocks(n_img, H, W, C, pool);
    const float inv_count = (float)(1.0 / ((double)n_img * H * W));
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* gi = (const bf16_t*)gin;
    const bf16_t* yy = (const bf16_t*)y;
    const double elems = (double)n_img * H * W * C;
    // both passes together: y read twice, gin read twice, dy written once
    ProfScope prof(pool ? "bn_act_bwd(reduce+apply)<pool>" : "bn_act_bwd(reduce+apply)", 0.0,
                   2.0 * elems * ((ym ? 1.25 : 2.0) + 2.0 * (pool ? 0.25 : 1.0) + 1.0), s);
    if (pool && rt) hipLaunchKernelGGL((bn_act_bwd_kernel<true, 0, true>), dim3(nb), dim3(256), 0, s, gi, yy, scale, shift, mean, rstd, sums, inv_count, partial, (bf16_t*)dy, rt, ym, g, d);
    else if (pool) hipLaunchKernelGGL((bn_act_bwd_kernel<true, 0, false>), dim3(nb), dim3(256), 0, s, gi, yy, scale, shift, mean, rstd, sums, inv_count, partial, (bf16_t*)dy, rt, ym, g, d);
    else hipLaunchKernelGGL((bn_act_bwd_kernel<false, 0, false>), dim3(nb), dim3(256), 0, s, gi, yy, scale, shift, mean, rstd, sums, inv_count, partial, (bf16_t*)dy, rt, ym, g, d);
    VAE_LAUNCH_CHECK("bn_act_bwd_kernel<0>");
    hipLaunchKernelGGL(sum_partials_kernel, dim3((2 * C + 31) / 32), dim3(1024), 0, s, partial, nb, 2 * C, 2 * C, sums, 1.0f);
    VAE_LAUNCH_CHECK("sum_partials_kernel");
    const int nb2 = elem_grid((long)n_img * H * W / (pool ? 4 : 1) * (C / 8));
    if (pool && rt) hipLaunchKernelGGL((bn_act_bwd_kernel<true, 1, true>), dim3(nb2), dim3(256), 0, s, gi, yy, scale, shift, mean, rstd, sums, inv_count, partial, (bf16_t*)dy, rt, ym, g, d);
    else if (pool) hipLaunchKernelGGL((bn_act_bwd_kernel<true, 1, false>), dim3(nb2), dim3(256), 0, s, gi, yy, scale, shift, mean, rstd, sums, inv_count, partial, (bf16_t*)dy, rt, ym, g, d);
    else hipLaunchKernelGGL((bn_act_bwd_kernel<false, 1, false>), dim3(nb2), dim3(256), 0, s, gi, yy, scale, shift, mean, rstd, sums, inv_count, partial, (bf16_t*)dy, rt, ym, g, d);
    VAE_LAUNCH_CHECK("bn_act_bwd_kernel<1>");
    return 0;
}

extern "C" int vae_head_fwd(const void* y, const float* scale, const float* shift, const float* w, const float* b,
                            float* recon, int n_img, int H, int W, int Cout, void* stream) {
    VAE_CHECK(y && scale && shift && w && b && recon, "vae_head_fwd: null pointer");
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31), "vae_head_fwd: bad geometry");
    VAE_CHECK(Cout >= 1 && Cout <= 4, "vae_head_fwd: out_channels=%d not supported (1..4)", Cout);
    const long npix = (long)n_img * H * W;
    // every lane first loads 16 + 8 * Cout coefficients: few long-lived workgroups here (more, shorter ones measured 17 % slower)
    const int grid = (int)((npix * 8 + 255) / 256 > 2048 ? 2048 : (npix * 8 + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("head_fwd_kernel", 2.0 * npix * 64 * Cout, npix * (128.0 + 4.0 * Cout), s);
#define GO(CO) hipLaunchKernelGGL(head_fwd_kernel<CO>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, scale, shift, w, b, recon, npix, H * W, pow2_shift(H * W))
    switch (Cout) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); }
#undef GO
    VAE_LAUNCH_CHECK("head_fwd_kernel");
    return 0;
}

// workgroups of the vector-ALU form of pass 0
static int head_bwd_pass0_blocks(long npix) {
    // pass 0 keeps 43 running sums per lane (164 VGPRs): three workgroups share a CU, so 768 is one full round of
    // resident workgroups on 256 CUs (1024 left a second round of 256 that ran alone)
    long b = (npix + 31) / 32;
    if (b < 1) b = 1;
    if (b > 768) b = 768;
    return (int)b;
}

extern "C" int vae_head_bwd_blocks(int n_img, int H, int W) {
    // rows of 256 floats (the narrowest row a caller multiplies by: (3 + 1) * 64) that hold either form's scratch
    const long npix = (long)n_img * H * W;
    const long a = (long)head_bwd_pass0_blocks(npix) * 7 * 64, b = head_sums_scratch_floats(npix);
    return (int)(((a > b ? a : b) + 255) / 256);
}

extern "C" int vae_head_bwd(const float* drecon, const float* recon, const void* y, const float* scale,
                            const float* shift, const float* mean, const float* rstd, const float* w, float* partial,
                            float* sums, float* sums_head, void* dy, int n_img, int H, int W, int Cout, void* stream) {
    VAE_CHECK(drecon && recon && y && scale && shift && mean && rstd && w && partial && sums && sums_head && dy, "vae_head_bwd: null pointer");
    VAE_CHECK(((uintptr_t)partial & 15) == 0, "vae_head_bwd: partial must be 16-byte aligned");
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31), "vae_head_bwd: bad geometry");
    VAE_CHECK(Cout >= 1 && Cout <= 4, "vae_head_bwd: out_channels=%d not supported (1..4)", Cout);
    const long npix = (long)n_img * H * W;
    const int nb = head_bwd_pass0_blocks(npix);
    const int ncols = (3 + Cout) * 64;
    const float inv_count = (float)(1.0 / (double)npix);
    hipStream_t s = (hipStream_t)stream;
    const bf16_t* yy = (const bf16_t*)y;
    ProfScope prof("head_bwd(reduce+apply)", 0.0, npix * (2 * 128.0 + 2 * 8.0 * Cout + 128.0), s);
#define GO(CO, PASS, NB) hipLaunchKernelGGL((head_bwd_kernel<CO, PASS>), dim3(NB), dim3(256), 0, s, drecon, recon, yy, scale, shift, mean, rstd, w, sums, inv_count, partial, (bf16_t*)dy, npix, H * W, pow2_shift(H * W))
    if (g_head_sums_mode == 0) {
        // every sum of pass 0 is a contraction over pixels: on the matrix pipe (bn_conv1_bwd.hip)
        if (int e = head_sums_mfma(drecon, recon, yy, scale, shift, mean, rstd, w, partial, sums, sums_head, npix, H * W, Cout, nullptr, s))
            return e;
    } else {
        switch (Cout) { case 1: GO(1, 0, nb); break; case 2: GO(2, 0, nb); break; case 3: GO(3, 0, nb); break; default: GO(4, 0, nb); }
        VAE_LAUNCH_CHECK("head_bwd_kernel<0>");
        // [d gamma | d beta] -> sums (also the coefficients of pass 1); [dW | db] -> sums_head
        hipLaunchKernelGGL(sum_partials_kernel, dim3(4), dim3(1024), 0, s, partial, nb, 128, ncols, sums, 1.0f);
        const int nh = Cout * 64 + Cout;
        hipLaunchKernelGGL(sum_partials_kernel, dim3((nh + 31) / 32), dim3(1024), 0, s, partial + 128, nb, nh, ncols, sums_head, 1.0f);
        VAE_LAUNCH_CHECK("sum_partials_kernel");
    }
    const int nb2 = (int)((npix + 31) / 32 > 2048 ? 2048 : (npix + 31) / 32);
    switch (Cout) { case 1: GO(1, 1, nb2); break; case 2: GO(2, 1, nb2); break; case 3: GO(3, 1, nb2); break; default: GO(4, 1, nb2); }
#undef GO
    VAE_LAUNCH_CHECK("head_bwd_kernel<1>");
    return 0;
}

extern "C" int vae_head_bwd_sums(const float* drecon, const float* recon, const void* y, const float* scale,
                                 const float* shift, const float* mean, const float* rstd, const float* w, float* partial,
                                 float* sums, float* sums_head, float* dl, int n_img, int H, int W, int Cout, void* stream) {
    VAE_CHECK(drecon && recon && y && scale && shift && mean && rstd && w && partial && sums && sums_head && dl, "vae_head_bwd_sums: null pointer");
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31), "vae_head_bwd_sums: bad geometry");
    VAE_CHECK(Cout >= 1 && Cout <= 4, "vae_head_bwd_sums: out_channels=%d not supported (1..4)", Cout);
    VAE_CHECK((((uintptr_t)partial | (uintptr_t)dl) & 15) == 0, "vae_head_bwd_sums: partial and dl must be 16-byte aligned");
    ProfScope prof("head_bwd_sums", 0.0, (double)n_img * H * W * (128.0 + 8.0 * Cout + 16.0), (hipStream_t)stream);
    return head_sums_mfma(drecon, recon, (const bf16_t*)y, scale, shift, mean, rstd, w, partial, sums, sums_head, (long)n_img * H * W,
                          H * W, Cout, dl, (hipStream_t)stream);
}

extern "C" int vae_tanh_bwd_rows(int rows) { return rows < 64 ? rows : 64; }

extern "C" int vae_tanh_bwd(const void* dx, const void* h, void* dpre, float* colpartial, int rows, int cols,
                            void* stream) {
    VAE_CHECK(dx && h && dpre && colpartial && rows > 0 && cols > 0 && cols % 8 == 0, "vae_tanh_bwd: bad arguments (cols must be a multiple of 8)");
    dim3 grid((cols / 8 + 255) / 256, vae_tanh_bwd_rows(rows));
    hipLaunchKernelGGL(tanh_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx, (const bf16_t*)h,
                       (bf16_t*)dpre, colpartial, rows, cols);
    VAE_LAUNCH_CHECK("tanh_bwd_kernel");
    return 0;
}

extern "C" int vae_nchw_to_nhwc8(const float* x, void* out, int n_img, int C, int H, int W, void* stream) {
    return vae_nchw_to_nhwc8_count(x, out, n_img, C, H, W, nullptr, 0, stream);
}

extern "C" int vae_nchw_to_nhwc8_count(const float* x, void* out, int n_img, int C, int H, int W, long long* counters,
                                       int ncount, void* stream) {
    VAE_CHECK(x && out && n_img > 0 && H > 0 && W > 0, "vae_nchw_to_nhwc8: bad arguments");
    VAE_CHECK(ncount >= 0 && ncount <= 256 && (((uintptr_t)counters) & 7) == 0, "vae_nchw_to_nhwc8_count: bad counters");
    VAE_CHECK(C >= 1 && C <= 8, "vae_nchw_to_nhwc8: in_channels=%d not supported (1..8)", C);
    const long npix = (long)n_img * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc8_kernel, dim3(elem_grid(npix)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out,
                       npix, C, H * W, counters, ncount);
    VAE_LAUNCH_CHECK("nchw_to_nhwc8_kernel");
    return 0;
}
