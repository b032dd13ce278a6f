// Shared device/host helpers for the gfx950 VAE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace vae {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);

#define VAE_CHECK(cond, ...)                                   \
    do {                                                       \
        if (!(cond)) {                                         \
            ::vae::set_error(__VA_ARGS__);                     \
            return 1;                                          \
        }                                                      \
    } while (0)

#define VAE_LAUNCH_CHECK(name)                                                     \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            ::vae::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                              \
        }                                                                          \
    } while (0)

// 256 bytes of zeros: the source of every padded / out-of-range row of a global_load_lds stage
// (one private copy per translation unit; no relocatable device code needed).
__device__ __attribute__((aligned(256))) static unsigned int g_zero_page[64];

// ---------------------------------------------------------------- bf16
typedef unsigned short bf16_t;  // raw bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) unsigned short ushort8v;
typedef __attribute__((ext_vector_type(4))) unsigned short ushort4v;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 and keeps NaN a NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
// two at once: the vector cast is ONE v_cvt_pk_bf16_f32 (packing two scalar casts costs four instructions)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const bf16x2v v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// Workgroup barrier for kernels that keep global->LDS copies (buffer_load ... lds) in flight across it.
// __syncthreads() carries a workgroup release fence, and because those copies write LDS the compiler
// turns the fence into s_waitcnt vmcnt(0): every barrier would drain the whole copy ring.  Here the caller
// has already waited (counted vmcnt) for exactly the copies the next phase reads; the wave's own LDS
// reads and writes are finished by lgkmcnt(0); the "memory" clobber keeps the compiler from moving
// memory operations across.
__device__ __forceinline__ void barrier_keep_copies() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------- reductions (wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block-wide sum; result valid in thread 0.  `red` is LDS scratch of >= blockDim/64 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    if (w == 0) {
        r = lane < nw ? red[lane] : 0.f;
        r = wave_sum(r);
    }
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------- Philox4x32-10
// Counter-based generator (Salmon et al., SC'11).  Stream layout used by every
// kernel here: element i of a draw is word i%4 of
//   philox(ctr = {i/4, step, stream, rank}, key = {seed_lo, seed_hi}).
struct PhiloxKey {
    uint32_t seed_lo, seed_hi, step, stream, rank;
};

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ void philox_block(const PhiloxKey& k, uint32_t blk, uint32_t out[4]) {
    out[0] = blk; out[1] = k.step; out[2] = k.stream; out[3] = k.rank;
    philox4x32_10(out, k.seed_lo, k.seed_hi);
}

// fp32 Box-Muller made only of individually (correctly) rounded mul/add/div/sqrt (no fma, no
// libm), so that host restatements reproduce it bit for bit.
__device__ __forceinline__ float log_f32_exact(float u) {
    const uint32_t bits = __float_as_uint(u);
    int e = (int)((bits >> 23) & 0xFF) - 127;
    float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421356237309504880f) { m = __fmul_rn(m, 0.5f); e += 1; }
    const float f = __fadd_rn(m, -1.0f);
    // s = f / (2 + f), correctly rounded: reciprocal, quotient, one correction -- the core of the IEEE division sequence
    // without its scaling and special-case steps (f is in [-0.30, 0.42], the divisor in [1.70, 2.42]).  The generator
    // calls this with the 2^24 values (k + 1) 2^-24 only, and over ALL of them it returns the bits of the full division
    // (vae_philox_fill mode 3 tabulates them; tests/test_gpu_kernels.py compares the table with numpy's division).
    const float den = __fadd_rn(2.0f, f);
    const float rc = __builtin_amdgcn_rcpf(den);
    const float q0 = __fmul_rn(f, rc);
    const float s = __builtin_fmaf(__builtin_fmaf(-den, q0, f), rc, q0);
    const float s2 = __fmul_rn(s, s);
    float p = (float)(2.0 / 11);
    p = __fadd_rn(__fmul_rn(p, s2), (float)(2.0 / 9));
    p = __fadd_rn(__fmul_rn(p, s2), (float)(2.0 / 7));
    p = __fadd_rn(__fmul_rn(p, s2), (float)(2.0 / 5));
    p = __fadd_rn(__fmul_rn(p, s2), (float)(2.0 / 3));
    p = __fadd_rn(__fmul_rn(p, s2), 2.0f);
    return __fadd_rn(__fmul_rn((float)e, 0.69314718055994530942f), __fmul_rn(p, s));
}

// Correctly rounded fp32 square root: sqrtf() compiles to v_sqrt_f32 plus the compiler's fix-up sequence (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt), i.e. the IEEE result, which is what numpy.sqrt returns on the
// host -- so the stream stays reproducible bit for bit.  (__fsqrt_rn is NOT this: it maps to the native, 1-ulp
// instruction.)  A third of the cost of the three-divide Newton iteration it replaces.
__device__ __forceinline__ float sqrt_f32_exact(float x) {
    return x > 0.f ? __builtin_sqrtf(x) : 0.f;
}

__device__ __forceinline__ void sincos_quarter(float r, float& sn, float& cs) {
    const bool swap = r > 0.5f;
    const float rr = swap ? __fadd_rn(1.0f, -r) : r;
    const float x = __fmul_rn(rr, 1.57079632679489661923f);
    const float x2 = __fmul_rn(x, x);
    float ps = (float)(1.0 / 362880);
    ps = __fadd_rn(__fmul_rn(ps, x2), (float)(-1.0 / 5040));
    ps = __fadd_rn(__fmul_rn(ps, x2), (float)(1.0 / 120));
    ps = __fadd_rn(__fmul_rn(ps, x2), (float)(-1.0 / 6));
    const float s = __fadd_rn(x, __fmul_rn(__fmul_rn(x, x2), ps));
    float pc = (float)(-1.0 / 3628800);
    pc = __fadd_rn(__fmul_rn(pc, x2), (float)(1.0 / 40320));
    pc = __fadd_rn(__fmul_rn(pc, x2), (float)(-1.0 / 720));
    pc = __fadd_rn(__fmul_rn(pc, x2), (float)(1.0 / 24));
    pc = __fadd_rn(__fmul_rn(pc, x2), -0.5f);
    const float c = __fadd_rn(1.0f, __fmul_rn(x2, pc));
    sn = swap ? c : s;
    cs = swap ? s : c;
}

__device__ __forceinline__ void normal_pair(uint32_t b0, uint32_t b1, float& z0, float& z1) {
    const float u1 = __fmul_rn(__fadd_rn((float)(b0 >> 8), 1.0f), 5.9604644775390625e-08f);  // 2^-24
    const uint32_t t = b1 >> 8;
    const uint32_t quad = t >> 22;
    const float r = __fmul_rn((float)(t & 0x3FFFFFu), 2.384185791015625e-07f);  // 2^-22
    float sn, cs;
    sincos_quarter(r, sn, cs);
    // angle = (quad + r) * pi/2:  cos = {cs,-sn,-cs,sn}[quad], sin = {sn,cs,-sn,-cs}[quad];
    // the signs are applied on the bit pattern so that no float select/negate folding is involved
    const uint32_t cbits = __float_as_uint((quad & 1u) ? sn : cs) ^ (((quad + 1u) & 2u) << 30);
    const uint32_t sbits = __float_as_uint((quad & 1u) ? cs : sn) ^ ((quad & 2u) << 30);
    const float c = __uint_as_float(cbits), s = __uint_as_float(sbits);
    const float rad = sqrt_f32_exact(__fmul_rn(-2.0f, log_f32_exact(u1)));
    z0 = __fmul_rn(rad, c);
    z1 = __fmul_rn(rad, s);
}

__device__ __forceinline__ void philox_normal4(const PhiloxKey& k, uint32_t blk, float z[4]) {
    uint32_t w[4];
    philox_block(k, blk, w);
    normal_pair(w[0], w[1], z[0], z[1]);
    normal_pair(w[2], w[3], z[2], z[3]);
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace vae
