// Activation helpers shared by the BatchNorm and GroupNorm kernels: 8-channel bf16 vectors (one 16-byte access),
// the dropout keep flags (injected mask or the Philox stream) and relu + dropout of a normalised vector.
#pragma once
#include "common.h"
#include "../../include/vae_hip.h"

namespace vae {

struct DropCfg {
    int enabled;
    uint32_t thr;
    float inv_keep;
    PhiloxKey key;
    const uint8_t* mask;   // optional injected keep-mask, same [pixel][C] layout as y
};

__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
    const uint4 u = *(const uint4*)p;
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
    *(uint4*)p = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}
// the same as one-pass streams: non-temporal (the line is not kept for this CU; measured +10 % on read-once /
// write-once kernels).  Use for tensors this kernel touches once; re-read tensors keep the plain forms.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4s;
__device__ __forceinline__ void load8s(const bf16_t* p, float v[8]) {
    const u32x4s u = __builtin_nontemporal_load((const u32x4s*)p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(u[i] << 16);
        v[2 * i + 1] = __uint_as_float(u[i] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ void store8s(bf16_t* p, const float v[8]) {
    const u32x4s u = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    __builtin_nontemporal_store(u, (u32x4s*)p);
}
__device__ __forceinline__ void load8f(const float* p, float v[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// keep flags of the 8 elements idx0..idx0+7 (idx0 % 8 == 0) of the dropout stream
__device__ __forceinline__ void keep8(const DropCfg& d, long idx0, bool keep[8]) {
    if (d.mask) {
        const uint2 m = *(const uint2*)(d.mask + idx0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            keep[i] = (m.x >> (8 * i)) & 0xFF;
            keep[4 + i] = (m.y >> (8 * i)) & 0xFF;
        }
    } else {
        // one Philox block serves 8 elements: element i takes the 16-bit half (i & 1) of word (i >> 1)
        uint32_t w[4];
        philox_block(d.key, (uint32_t)(idx0 >> 3), w);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            keep[2 * i] = (w[i] & 0xFFFFu) >= d.thr;
            keep[2 * i + 1] = (w[i] >> 16) >= d.thr;
        }
    }
}

// d = dropout(relu(y*sc + sh)) for 8 channels; pos[i] = relu active, kp[i] = kept
__device__ __forceinline__ void act8(const float y[8], const float sc[8], const float sh[8], const DropCfg& d,
                                     long idx0, float out[8], bool pos[8], bool kp[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) kp[i] = true;
    if (d.enabled) keep8(d, idx0, kp);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float a = y[i] * sc[i] + sh[i];
        pos[i] = a > 0.f;
        float v = pos[i] ? a : 0.f;
        if (d.enabled) v = kp[i] ? v * d.inv_keep : 0.f;
        out[i] = v;
    }
}

static inline DropCfg make_drop(double p_drop, const void* mask, unsigned long long seed, unsigned step, unsigned stream_id,
                         unsigned rank) {
    DropCfg d = {};
    d.enabled = p_drop > 0.0;
    d.thr = vae_dropout_threshold(p_drop);
    d.inv_keep = (float)(1.0 / (1.0 - p_drop));
    d.key.seed_lo = (uint32_t)seed; d.key.seed_hi = (uint32_t)(seed >> 32);
    d.key.step = step; d.key.stream = stream_id; d.key.rank = rank;
    d.mask = (const uint8_t*)mask;
    return d;
}

static inline int check_geom(const char* who, int n_img, int H, int W, int C, int pool) {
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && C > 0, "%s: bad geometry n=%d H=%d W=%d C=%d", who, n_img, H, W, C);
    VAE_CHECK((long)n_img * H * W < (1L << 31), "%s: more than 2^31 pixels", who);
    VAE_CHECK(C % 8 == 0 && 256 % (C / 8) == 0 && C <= 2048, "%s: C=%d must be 8 * a divisor of 256", who, C);
    VAE_CHECK(!pool || (H % 2 == 0 && W % 2 == 0), "%s: pooling needs even H, W (got %dx%d)", who, H, W);
    return 0;
}

}  // namespace vae
