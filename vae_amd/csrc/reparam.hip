// Reparameterisation sample fused with the KL term, forward and backward, plus the raw
// Philox stream kernels used by the RNG parity tests.
//
//   z   = mu + exp(0.5*logvar) * eps            eps ~ N(0,1) from Philox4x32-10 (or injected)
//   KL  = -0.5 * sum(1 + logvar - mu^2 - exp(logvar))
//   dmu = dz + dmu_ext + kl_scale * mu
//   dlv = dz * eps * 0.5 * exp(0.5*logvar) + dlv_ext + kl_scale * 0.5 * (exp(logvar) - 1)
//
// HBM-bound; forward moves 12 algorithmic bytes per latent element (mu, logvar in, z out),
// eps is regenerated from the counter in the backward instead of being stored.
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

__device__ __forceinline__ PhiloxKey make_key(unsigned long long seed, unsigned step,
                                              unsigned stream, unsigned rank) {
    PhiloxKey k;
    k.seed_lo = (uint32_t)seed;
    k.seed_hi = (uint32_t)(seed >> 32);
    k.step = step;
    k.stream = stream;
    k.rank = rank;
    return k;
}

// SLABS: mu | logvar are not read but formed here from the split-K slabs of the fc_mu | fc_logvar GEMM
// (slab[s][row][2L], bias[2L]; bias first, then the slabs in order -- the sum the GEMM's own finishing kernel forms)
// and written out: one dependent launch fewer between the encoder and the decoder.
template <bool SLABS>
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(
    float* __restrict__ mu, float* __restrict__ logvar, const float* __restrict__ eps_in,
    float* __restrict__ z_f32, bf16_t* __restrict__ z_bf16, float* __restrict__ eps_out,
    float* __restrict__ kl_partial, long n, unsigned long long seed, unsigned step, unsigned rank,
    const float* __restrict__ slab, int nsplit, const float* __restrict__ bias, int L) {
    __shared__ float red[8];
    const PhiloxKey key = make_key(seed, step, VAE_STREAM_EPS, rank);
    const long nblk = (n + 3) >> 2;
    float acc = 0.f;
    for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
         b += (long)gridDim.x * blockDim.x) {
        const long i0 = b << 2;
        float m[4], lv[4], e[4], z[4];
        const bool full = i0 + 4 <= n;
        if (SLABS) {           // n is a multiple of L, L of 4: whole float4s inside one row
            const long row = i0 / L;
            const int col = (int)(i0 - row * L);
            const long e = row * 2 * L + col, mn = (n / L) * 2 * L;
            float4 a = *(const float4*)(bias + col), b4 = *(const float4*)(bias + L + col);
            int s = 0;
            for (; s + 3 < nsplit; s += 4) {       // four slabs of both halves requested at once, added in slab order
                float4 u[4], v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u[i] = *(const float4*)(slab + (long)(s + i) * mn + e);
                    v[i] = *(const float4*)(slab + (long)(s + i) * mn + e + L);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a.x += u[i].x; a.y += u[i].y; a.z += u[i].z; a.w += u[i].w;
                    b4.x += v[i].x; b4.y += v[i].y; b4.z += v[i].z; b4.w += v[i].w;
                }
            }
            for (; s < nsplit; ++s) {
                const float4 u = *(const float4*)(slab + (long)s * mn + e), v = *(const float4*)(slab + (long)s * mn + e + L);
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                b4.x += v.x; b4.y += v.y; b4.z += v.z; b4.w += v.w;
            }
            *(float4*)(mu + i0) = a;
            *(float4*)(logvar + i0) = b4;
            m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
            lv[0] = b4.x; lv[1] = b4.y; lv[2] = b4.z; lv[3] = b4.w;
        } else if (full) {
            const float4 m4 = *(const float4*)(mu + i0);
            const float4 l4 = *(const float4*)(logvar + i0);
            m[0] = m4.x; m[1] = m4.y; m[2] = m4.z; m[3] = m4.w;
            lv[0] = l4.x; lv[1] = l4.y; lv[2] = l4.z; lv[3] = l4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m[j] = i0 + j < n ? mu[i0 + j] : 0.f;
                lv[j] = i0 + j < n ? logvar[i0 + j] : 0.f;
            }
        }
        if (eps_in) {
            if (full) {
                const float4 e4 = *(const float4*)(eps_in + i0);
                e[0] = e4.x; e[1] = e4.y; e[2] = e4.z; e[3] = e4.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) e[j] = i0 + j < n ? eps_in[i0 + j] : 0.f;
            }
        } else {
            philox_normal4(key, (uint32_t)b, e);
        }
        // one exponential per element: exp(lv) = sd * sd (two roundings instead of one, ~1e-7 relative), and that one
        // from the hardware's exp2 (__expf: a multiply and v_exp_f32, ~2e-7 relative; the accurate expf is ~12 vector
        // instructions and this kernel is ALU-bound beside the Philox stream).  Results are compared at 1e-5.
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sd = __expf(0.5f * lv[j]);
            z[j] = m[j] + sd * e[j];
            t[j] = 1.f + lv[j] - m[j] * m[j] - sd * sd;
        }
        if (full) {            // whole block: no per-element bounds checks (same order of the four additions as below)
            acc += t[0]; acc += t[1]; acc += t[2]; acc += t[3];
            if (z_f32) *(float4*)(z_f32 + i0) = make_float4(z[0], z[1], z[2], z[3]);
            if (z_bf16) *(uint2*)(z_bf16 + i0) = make_uint2(pack2bf(z[0], z[1]), pack2bf(z[2], z[3]));
            if (eps_out) *(float4*)(eps_out + i0) = make_float4(e[0], e[1], e[2], e[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < n) {
                    acc += t[j];
                    if (z_f32) z_f32[i0 + j] = z[j];
                    if (z_bf16) z_bf16[i0 + j] = f2bf(z[j]);
                    if (eps_out) eps_out[i0 + j] = e[j];
                }
        }
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) kl_partial[blockIdx.x] = -0.5f * s;
}

// dcat (bf16, optional) holds [row][0:L] = dmu, [row][L:2L] = dlogvar, the operand layout of
// the fused mu|logvar head's dgrad / wgrad GEMMs.
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ mu, const float* __restrict__ logvar,
    const float* __restrict__ eps_in, const float* __restrict__ dmu_ext,
    const float* __restrict__ dlv_ext, float kl_scale, float* __restrict__ dmu,
    float* __restrict__ dlv, bf16_t* __restrict__ dcat, long n, int L, unsigned long long seed,
    unsigned step, unsigned rank, int nsplit) {
    // nsplit > 0: dz points at the nsplit split-K slabs ([s][n]) of the GEMM that produces it; they are added here, in order
    const PhiloxKey key = make_key(seed, step, VAE_STREAM_EPS, rank);
    const long nblk = (n + 3) >> 2;
    for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
         b += (long)gridDim.x * blockDim.x) {
        const long i0 = b << 2;
        float e[4];
        if (eps_in) {
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = i0 + j < n ? eps_in[i0 + j] : 0.f;
        } else {
            philox_normal4(key, (uint32_t)b, e);
        }
        float gsum[4] = {0.f, 0.f, 0.f, 0.f};
        if (dz && nsplit > 0 && i0 + 4 <= n) {       // (n is a multiple of 4 whenever slabs are used: checked on the host)
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            int s = 0;
            for (; s + 3 < nsplit; s += 4) {
                float4 u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = *(const float4*)(dz + (long)(s + i) * n + i0);
#pragma unroll
                for (int i = 0; i < 4; ++i) { a.x += u[i].x; a.y += u[i].y; a.z += u[i].z; a.w += u[i].w; }
            }
            for (; s < nsplit; ++s) {
                const float4 u = *(const float4*)(dz + (long)s * n + i0);
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            }
            gsum[0] = a.x; gsum[1] = a.y; gsum[2] = a.z; gsum[3] = a.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long i = i0 + j;
            if (i >= n) break;
            float g = 0.f;
            if (dz) g = nsplit > 0 ? gsum[j] : dz[i];
            const float lv = logvar[i];
            const float sd = __expf(0.5f * lv);
            float gm = g + kl_scale * mu[i];
            float gl = g * e[j] * 0.5f * sd + kl_scale * 0.5f * (sd * sd - 1.f);
            if (dmu_ext) gm += dmu_ext[i];
            if (dlv_ext) gl += dlv_ext[i];
            if (dmu) dmu[i] = gm;
            if (dlv) dlv[i] = gl;
            if (dcat) {
                const long row = i / L;
                const int col = (int)(i - row * L);
                dcat[row * 2 * L + col] = f2bf(gm);
                dcat[row * 2 * L + L + col] = f2bf(gl);
            }
        }
    }
}

// MODE 0: raw uint32 words, 1: normals (fp32), 2: keep-mask bytes (8 per Philox block: element i of the
// stream takes the 16-bit half (i & 1) of word (i >> 1) & 3 of block i >> 3, kept iff half >= thr)
template <int MODE>
__global__ __launch_bounds__(256) void philox_fill_kernel(void* __restrict__ out, long n,
                                                          PhiloxKey key, uint32_t thr) {
    const long nblk = MODE == 2 ? (n + 7) >> 3 : (n + 3) >> 2;
    for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
         b += (long)gridDim.x * blockDim.x) {
        const long i0 = MODE == 2 ? b << 3 : b << 2;
        if (MODE == 2) {
            uint32_t w[4];
            philox_block(key, (uint32_t)b, w);
            for (int j = 0; j < 8 && i0 + j < n; ++j)
                ((uint8_t*)out)[i0 + j] = ((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu) >= thr ? 1 : 0;
        } else if (MODE == 1) {
            float z[4];
            philox_normal4(key, (uint32_t)b, z);
            for (int j = 0; j < 4 && i0 + j < n; ++j) ((float*)out)[i0 + j] = z[j];
        } else if (MODE == 3) {
            // the generator's logarithm over its whole domain: element i = ln((i + 1) 2^-24), the uniform of 24-bit draw i
            for (int j = 0; j < 4 && i0 + j < n; ++j)
                ((float*)out)[i0 + j] = log_f32_exact(__fmul_rn(__fadd_rn((float)(uint32_t)(i0 + j), 1.0f), 5.9604644775390625e-08f));
        } else {
            uint32_t w[4];
            philox_block(key, (uint32_t)b, w);
            for (int j = 0; j < 4 && i0 + j < n; ++j) {
                ((uint32_t*)out)[i0 + j] = w[j];
            }
        }
    }
}

}  // namespace vae

using namespace vae;

extern "C" int vae_reparam_grid(long n_elems) {
    long blocks = ((n_elems + 3) / 4 + 255) / 256;      // one Philox block (four elements) per lane
    if (blocks < 1) blocks = 1;
    if (blocks > (1L << 20)) blocks = 1L << 20;         // grid-stride beyond 2^30 elements
    return (int)blocks;
}

extern "C" int vae_reparam_kl_fwd(const float* mu, const float* logvar, const float* eps_in,
                                  float* z_f32, void* z_bf16, float* eps_out, float* kl_partial,
                                  long n_elems, unsigned long long seed, unsigned step,
                                  unsigned rank, void* stream) {
    VAE_CHECK(mu && logvar && kl_partial, "vae_reparam_kl_fwd: null pointer");
    VAE_CHECK(n_elems > 0, "vae_reparam_kl_fwd: n_elems must be > 0 (got %ld)", n_elems);
    VAE_CHECK((((uintptr_t)mu | (uintptr_t)logvar | (uintptr_t)z_f32 | (uintptr_t)eps_out) & 15) == 0 &&
                  ((uintptr_t)z_bf16 & 7) == 0,
              "vae_reparam_kl_fwd: pointers must be 16-byte aligned");
    VAE_CHECK(n_elems < (1L << 34), "vae_reparam_kl_fwd: n_elems too large for a 32-bit block counter");
    const int grid = vae_reparam_grid(n_elems);
    ProfScope prof("reparam_kl_fwd_kernel", 0.0, n_elems * (8.0 + (eps_in ? 4.0 : 0.0) + (z_f32 ? 4.0 : 0.0) + (z_bf16 ? 2.0 : 0.0) + (eps_out ? 4.0 : 0.0)), (hipStream_t)stream);
    hipLaunchKernelGGL(reparam_kl_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)mu,
                       (float*)logvar, eps_in, z_f32, (bf16_t*)z_bf16, eps_out, kl_partial, n_elems, seed,
                       step, rank, (const float*)nullptr, 0, (const float*)nullptr, 0);
    VAE_LAUNCH_CHECK("reparam_kl_fwd_kernel");
    return 0;
}

extern "C" int vae_reparam_kl_fwd_slabs(const float* slab, int nsplit, const float* bias, float* mu, float* logvar,
                                        const float* eps_in, float* z_f32, void* z_bf16, float* eps_out, float* kl_partial,
                                        long n_elems, int latent, unsigned long long seed, unsigned step, unsigned rank,
                                        void* stream) {
    VAE_CHECK(slab && bias && mu && logvar && kl_partial && nsplit >= 1, "vae_reparam_kl_fwd_slabs: null pointer or no slabs");
    VAE_CHECK(n_elems > 0 && latent > 0 && latent % 4 == 0 && n_elems % latent == 0,
              "vae_reparam_kl_fwd_slabs: n_elems (%ld) must be a positive multiple of latent (%d), latent of 4", n_elems, latent);
    VAE_CHECK((((uintptr_t)slab | (uintptr_t)bias | (uintptr_t)mu | (uintptr_t)logvar | (uintptr_t)z_f32 | (uintptr_t)eps_out) & 15) == 0 &&
                  ((uintptr_t)z_bf16 & 7) == 0,
              "vae_reparam_kl_fwd_slabs: pointers must be 16-byte aligned");
    VAE_CHECK(n_elems < (1L << 34), "vae_reparam_kl_fwd_slabs: n_elems too large for a 32-bit block counter");
    const int grid = vae_reparam_grid(n_elems);
    ProfScope prof("reparam_kl_fwd_kernel", 0.0, n_elems * (8.0 * (nsplit + 1) + (eps_in ? 4.0 : 0.0) + (z_f32 ? 4.0 : 0.0) + (z_bf16 ? 2.0 : 0.0) + (eps_out ? 4.0 : 0.0)), (hipStream_t)stream);
    hipLaunchKernelGGL(reparam_kl_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, mu, logvar, eps_in, z_f32,
                       (bf16_t*)z_bf16, eps_out, kl_partial, n_elems, seed, step, rank, slab, nsplit, bias, latent);
    VAE_LAUNCH_CHECK("reparam_kl_fwd_kernel<slabs>");
    return 0;
}

extern "C" int vae_reparam_kl_bwd(const float* dz, const float* mu, const float* logvar,
                                  const float* eps_in, const float* dmu_ext, const float* dlv_ext,
                                  float kl_scale, float* dmu, float* dlv, void* dcat_bf16,
                                  long n_elems, int latent, unsigned long long seed, unsigned step,
                                  unsigned rank, void* stream) {
    return vae_reparam_kl_bwd_slabs(dz, 0, mu, logvar, eps_in, dmu_ext, dlv_ext, kl_scale, dmu, dlv, dcat_bf16, n_elems, latent,
                                    seed, step, rank, stream);
}

extern "C" int vae_reparam_kl_bwd_slabs(const float* dz, int nsplit, const float* mu, const float* logvar,
                                        const float* eps_in, const float* dmu_ext, const float* dlv_ext,
                                        float kl_scale, float* dmu, float* dlv, void* dcat_bf16,
                                        long n_elems, int latent, unsigned long long seed, unsigned step,
                                        unsigned rank, void* stream) {
    VAE_CHECK(mu && logvar, "vae_reparam_kl_bwd: null pointer");
    VAE_CHECK(nsplit >= 0 && (nsplit == 0 || dz), "vae_reparam_kl_bwd_slabs: nsplit=%d slabs need dz", nsplit);
    VAE_CHECK(nsplit == 0 || (n_elems % 4 == 0 && ((uintptr_t)dz & 15) == 0), "vae_reparam_kl_bwd_slabs: slabs need n_elems a multiple of 4 and 16-byte alignment");
    VAE_CHECK(n_elems > 0 && latent > 0 && n_elems % latent == 0,
              "vae_reparam_kl_bwd: n_elems (%ld) must be a positive multiple of latent (%d)",
              n_elems, latent);
    const int grid = vae_reparam_grid(n_elems);
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dz, mu,
                       logvar, eps_in, dmu_ext, dlv_ext, kl_scale, dmu, dlv, (bf16_t*)dcat_bf16,
                       n_elems, latent, seed, step, rank, nsplit);
    VAE_LAUNCH_CHECK("reparam_kl_bwd_kernel");
    return 0;
}

extern "C" int vae_philox_fill(void* out, long n, int mode, double p_drop, unsigned long long seed,
                               unsigned step, unsigned stream_id, unsigned rank, void* stream) {
    VAE_CHECK(out && n > 0, "vae_philox_fill: bad arguments");
    VAE_CHECK(mode >= 0 && mode <= 3, "vae_philox_fill: unknown mode %d", mode);
    VAE_CHECK(mode != 3 || n <= (1L << 24), "vae_philox_fill: mode 3 tabulates at most 2^24 values");
    PhiloxKey key;
    key.seed_lo = (uint32_t)seed; key.seed_hi = (uint32_t)(seed >> 32);
    key.step = step; key.stream = stream_id; key.rank = rank;
    const uint32_t thr = vae_dropout_threshold(p_drop);
    const int grid = vae_reparam_grid(n);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 0) hipLaunchKernelGGL(philox_fill_kernel<0>, dim3(grid), dim3(256), 0, s, out, n, key, thr);
    else if (mode == 1) hipLaunchKernelGGL(philox_fill_kernel<1>, dim3(grid), dim3(256), 0, s, out, n, key, thr);
    else if (mode == 3) hipLaunchKernelGGL(philox_fill_kernel<3>, dim3(grid), dim3(256), 0, s, out, n, key, thr);
    else hipLaunchKernelGGL(philox_fill_kernel<2>, dim3(grid), dim3(256), 0, s, out, n, key, thr);
    VAE_LAUNCH_CHECK("philox_fill_kernel");
    return 0;
}

extern "C" unsigned vae_dropout_threshold(double p_drop) {
    double t = p_drop * 65536.0;               // 16-bit draws: P(keep) = 1 - thr / 2^16
    double c = (double)(unsigned long long)t;
    if (c < t) c += 1.0;
    if (c > 65536.0) c = 65536.0;
    if (c < 0) c = 0;
    return (unsigned)c;
}
