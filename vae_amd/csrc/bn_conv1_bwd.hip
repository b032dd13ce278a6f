// Backward of the network's first layer group  Conv2d(Cin <= 3, 64, 3, padding=1) -> BatchNorm2d -> ReLU  in ONE pass
// over the two large tensors (vae/models/vae.py:24-31, the first conv of the first DoubleConv).
//
// No layer in front of the first conv needs an input gradient, so the gradient g of the conv output is only ever
// used in sums over all pixels -- and g is affine in quantities a single pass can accumulate.  With
//   d  = gin where the ReLU passed, 0 elsewhere          (gin = gradient of the activation)
//   xh = (y - mean) * rstd                                (y = stored conv output)
//   g  = scale * (d - mean_p(d) - xh * mean_p(d * xh))    (BatchNorm backward, scale = gamma * rstd)
// the weight gradient  dW[co][j] = sum_p g[p][co] * X[p][j]  (X[p][j]: input pixel p shifted by tap j / channel) is
//   dW = scale * (T1 - mean(d) * T2 - mean(d * xh) * T3),
//   T1[co][j] = sum_p d[p][co] X[p][j],   T2[j] = sum_p X[p][j],   T3[co][j] = rstd * (sum_p y[p][co] X[p][j] - mean * T2[j]).
// One kernel reads gin and y once and accumulates everything on the matrix pipe -- T1, sum y X, T2 and, for d gamma
// and d beta, sum d (a column of ones in X) and sum d * y (the diagonal of d^T y): per 64-pixel tile the bf16 values
// [d | y] (128 columns, d is exact in bf16: it is gin or 0) and the 9 x Cin shifted input values of each pixel go to
// LDS, and the K = pixel contraction runs as 16x16x32 MFMAs on transposing LDS reads; T2 comes from a constant
// all-ones operand built in registers.  Neither g
// nor any other full-size tensor is written: against the separate BatchNorm-backward passes plus the weight-gradient
// GEMM this drops one write and two reads of the [pixels][64] tensor and three launches.
// Every workgroup leaves one row of partial results; a second kernel (one workgroup per output channel) folds the rows
// in fp64 and fixed order and forms dW, d gamma and d beta.
// The same contraction with other tiles gives the per-channel sums of the 1x1 head's backward (MODE 1 below).
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

int g_conv1_bwd_mode = 0;

constexpr int C1_BK = 64;                                   // pixels per tile
constexpr int C1_ROWA = 256, C1_ROWX = 64;                  // bytes per LDS row: 64 d + 64 y | 32 shifted-input columns
constexpr int C1_A_BYTES = C1_BK * C1_ROWA, C1_X_BYTES = C1_BK * C1_ROWX, C1_STAGE = C1_A_BYTES + C1_X_BYTES;
constexpr int C1_COLS = 32, C1_TROWS = 129;                 // T rows: 64 x d, 64 x y, 1 x ones
constexpr int C1_ROWF = C1_TROWS * C1_COLS + 64;            // floats per workgroup: T | sum d*y (64); sum d is column 31 of T
constexpr int C1_MAX_BLOCKS = 768;                         // three resident workgroups per CU

struct Conv1Bwd {
    const bf16_t* gin;
    const bf16_t* y;
    const bf16_t* x8;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* rstd;
    float* part;
    int npix, H, W, wshift, hshift, Cin, ntile;
    // head form (MODE 1): the loss gradient and the reconstruction, NCHW fp32 with hw pixels per plane
    const float* drecon;
    const float* recon;
    int hw, hwshift;
    float* dl_out;        // optional: the logit gradients dl [pixel][4] (for the fused backward of the layer in front)
    // bits form (MODE 2): the conv output was never stored.  One bit per element says where the ReLU passed; the tile's
    // y is rebuilt from the shifted-input tile and the conv's own weight and bias (fp32, parameter layout)
    const unsigned char* relu_bits;   // [pixel][8]
    const float* w;                   // [64][Cin][3][3]
    const float* bias;                // [64]
};

// swizzles of the 16-byte chunk index: the 8 rows {b..b+3, b+8..b+11} one half of a transposing read touches land on
// 8 distinct 32-byte slots of the 256-byte bank line
__device__ __forceinline__ int c1_swz_a(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }
__device__ __forceinline__ int c1_swz_x(int row) { return ((row >> 3) & 1) << 1; }

// lanes 16g..16g+15 fetch rows k0+8g .. k0+8g+7 of columns col0..col0+15 transposed: lane (l & 15) gets column
// col0 + (l & 15), 8 consecutive k.  Inline asm: the caller waits with c1_wait before the first use.
template <bool XT>
__device__ __forceinline__ bf16x8 c1_frag(unsigned base, int k0, int col0, int lane) {
    constexpr int ROWB = XT ? C1_ROWX : C1_ROWA;
    const int g = lane >> 4, t = lane & 15;
    const int row = k0 + 8 * g + (t >> 2);
    const int off = (col0 + 4 * (t & 3)) * 2;
    const int chunk = off >> 4, within = off & 15;
    const int s0 = XT ? c1_swz_x(row) : c1_swz_a(row), s1 = XT ? c1_swz_x(row + 4) : c1_swz_a(row + 4);
    const unsigned a = base + row * ROWB + (((chunk ^ s0) << 4) | within);
    const unsigned b = base + (row + 4) * ROWB + (((chunk ^ s1) << 4) | within);
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b) : "memory");
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void c1_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f) : : "memory"); }

// MODE 0: the first conv (NC = input channels).  MODE 1: the sums of the 1x1 head's backward (NC = output channels), the
// same contraction over pixels with other tiles: [pos * y | pos] (pos = 1.0 where the ReLU passed) against the logit
// gradients dl[p][o] = drecon * recon * (1 - recon) split into two bf16 terms (columns o and 4 + o: products exact, the
// split is good to 2^-17).  Everything the head's backward sums over pixels follows from P2[o][c] = sum_p dl (pos y) and
// P1[o][c] = sum_p dl pos:  dW[o][c] = scale P2 + shift P1  (a = relu(scale y + shift)),  sum g = sum_o w[o][c] P1,
// sum g y = sum_o w[o][c] P2  (g = pos * sum_o w[o][c] dl[o]),  db[o] = sum_p dl  (the all-ones row).
// MODE 2: MODE 0 without the stored conv output (see Conv1Bwd): d = gin where the forward's ReLU bit is set, and the y
// half of the tile is y = bf16(X W'^T + bias) from the tile's shifted-input rows -- one 16x16x32 matrix instruction per 16
// pixels x 16 channels (K = the 32 columns of the X tile; W' is the weight in that column order, bf16 as the forward
// used it).  The forward summed the same products in another order, so a rebuilt y may differ from the forward's in
// its last bf16 bit; y only enters sums over all pixels here (the ReLU decision itself comes from the bits).
template <int MODE, int NC>
__global__ __launch_bounds__(256, 3) void bn_bwd_conv1_kernel(const Conv1Bwd p) {
    constexpr int CIN = NC;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * C1_STAGE];
    __shared__ __attribute__((aligned(16))) unsigned char wlds[MODE == 2 ? 64 * 64 : 16];
    __shared__ __attribute__((aligned(16))) float bias_l[MODE == 2 ? 64 : 4];
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = tid & 7, pl = tid >> 3;                  // channel group (8 channels) and pixel lane (0..31) of the load phase
    const int c0 = cg * 8;

    // columns 9 * Cin .. 31 of the shifted-input tiles are never written: clear everything once
    for (int i = tid; i < 2 * C1_STAGE / 16; i += 256) ((uint4*)smem)[i] = make_uint4(0u, 0u, 0u, 0u);

    __syncthreads();
    // column 31 is 1.0 in every row (nine taps x Cin <= 27 columns are rewritten per tile, this one never): T1[c][31] = sum d
    if (tid < 128) {
        const int row = tid & 63;
        *(bf16_t*)(smem + (tid >> 6) * C1_STAGE + C1_A_BYTES + row * C1_ROWX + (((3 ^ c1_swz_x(row)) << 4) | 14)) = (bf16_t)0x3F80;
    }
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = MODE == 2 ? 0.f : p.scale[c0 + i]; sh[i] = MODE == 2 ? 0.f : p.shift[c0 + i]; }
    if (MODE == 2) {
        // W'[co][k], k = tap * Cin + ci (the X tile's column order), zero from 9 * Cin on (column 31 of X is all ones)
        for (int i = tid; i < 64 * 32; i += 256) {
            const int co = i >> 5, k = i & 31, tap = k / CIN, ci = k - tap * CIN;
            ((bf16_t*)wlds)[i] = k < 9 * CIN ? f2bf(p.w[(co * CIN + ci) * 9 + tap]) : (bf16_t)0;
        }
        if (tid < 64) bias_l[tid] = p.bias ? p.bias[tid] : 0.f;
    }

    // registers of the tile in flight: two pixels x (gin, y) of this lane's channel group, up to three (pixel, tap) pairs
    // (a second set in flight, so that memory is never idle, measured slower in every form tried: 40-46 vs 32-38 us)
    struct InFlight {
        uint4 g[2], y[2];
        unsigned bits[2];                   // MODE 2: the ReLU bits of this lane's 8 channels
        uint2 x[3];
        float hr[4], hd[4];                 // MODE 1: recon and drecon of this lane's pixel (lanes 0..63)
    };
    InFlight r0;
    auto fetch = [&](InFlight& r, int tile) {
        uint4(&rg)[2] = r.g;
        uint4(&ry)[2] = r.y;
        uint2(&rx)[3] = r.x;
        const int k0 = tile * C1_BK;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = k0 + pl + 32 * h;
            rg[h] = ry[h] = make_uint4(0u, 0u, 0u, 0u);
            r.bits[h] = 0u;
            if (k < p.npix) {
                if (MODE != 1) rg[h] = *(const uint4*)(p.gin + (long)k * 64 + c0);
                if (MODE == 2) r.bits[h] = p.relu_bits[(long)k * 8 + cg];
                else ry[h] = *(const uint4*)(p.y + (long)k * 64 + c0);
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int o = 0; o < 4; ++o) r.hr[o] = r.hd[o] = 0.f;
            const int k = k0 + tid;
            if (tid < C1_BK && k < p.npix) {
                const unsigned up = (unsigned)k;
                const unsigned img = p.hwshift >= 0 ? up >> p.hwshift : up / (unsigned)p.hw, pp = up - img * (unsigned)p.hw;
#pragma unroll
                for (int o = 0; o < NC; ++o) {
                    const long e = ((long)img * NC + o) * p.hw + pp;
                    r.hr[o] = p.recon[e];
                    r.hd[o] = p.drecon[e];
                }
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 256 * q;                  // (pixel, tap) pair: 64 x 9 = 576 of them
            rx[q] = make_uint2(0u, 0u);
            if (idx < C1_BK * 9) {
                const int px = idx / 9, tap = idx - px * 9;
                const int k = k0 + px;
                int h, w;
                if (p.wshift >= 0 && p.hshift >= 0) { w = k & (p.W - 1); h = (k >> p.wshift) & (p.H - 1); }
                else { w = k % p.W; h = (k / p.W) % p.H; }
                const int dh = tap / 3 - 1, dw = tap - (tap / 3) * 3 - 1;
                if (k < p.npix && (unsigned)(h + dh) < (unsigned)p.H && (unsigned)(w + dw) < (unsigned)p.W)
                    rx[q] = *(const uint2*)(p.x8 + (long)(k + dh * p.W + dw) * 8);
            }
        }
    };
    // the fetched registers -> LDS stage `buf`: d = gin where the ReLU passed (the forward's fp32 predicate), y as it is
    auto deposit = [&](const InFlight& r, int buf, int tile) {
        const uint4(&rg)[2] = r.g;
        const uint4(&ry)[2] = r.y;
        const uint2(&rx)[3] = r.x;
        unsigned char* abase = smem + buf * C1_STAGE;
        unsigned char* xbase = abase + C1_A_BYTES;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = pl + 32 * h;
            const uint32_t gw[4] = {rg[h].x, rg[h].y, rg[h].z, rg[h].w}, yw[4] = {ry[h].x, ry[h].y, ry[h].z, ry[h].w};
            uint32_t dwd[4], gw1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t keep = 0u;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * j + e;
                    const float v = __uint_as_float(e ? (yw[j] & 0xFFFF0000u) : (yw[j] << 16));
                    const bool pass = MODE == 2 ? ((r.bits[h] >> i) & 1u) != 0u : v * sc[i] + sh[i] > 0.f;
                    if (pass) keep |= e ? 0xFFFF0000u : 0x0000FFFFu;
                }
                dwd[j] = MODE != 1 ? (gw[j] & keep) : (yw[j] & keep);
                if (MODE == 1) gw1[j] = 0x3F803F80u & keep;
            }
            const int s = c1_swz_a(row);
            *(uint4*)(abase + row * C1_ROWA + ((cg ^ s) << 4)) = make_uint4(dwd[0], dwd[1], dwd[2], dwd[3]);
            if (MODE != 2)     // (MODE 2: make_y fills the y half once the X tile is complete)
                *(uint4*)(abase + row * C1_ROWA + (((8 + cg) ^ s) << 4)) = MODE == 0 ? ry[h] : make_uint4(gw1[0], gw1[1], gw1[2], gw1[3]);
        }
        if (MODE == 1) {
            if (tid < C1_BK) {
                uint32_t hi[4], lo[4];
                float dlv[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float dl = r.hd[o] * r.hr[o] * (1.f - r.hr[o]);
                    dlv[o] = dl;
                    const bf16_t h16 = f2bf(dl);
                    hi[o] = h16;
                    lo[o] = f2bf(dl - bf2f(h16));
                }
                if (p.dl_out && tile * C1_BK + tid < p.npix)
                    *(float4*)(p.dl_out + ((long)tile * C1_BK + tid) * 4) = make_float4(dlv[0], dlv[1], dlv[2], dlv[3]);
                *(uint4*)(xbase + tid * C1_ROWX + (c1_swz_x(tid) << 4)) =
                    make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = tid + 256 * q;
            if (idx < C1_BK * 9) {
                const int px = idx / 9, tap = idx - px * 9;
                const uint32_t xw[2] = {rx[q].x, rx[q].y};
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    const int off = (tap * CIN + ci) * 2;
                    const bf16_t v = (bf16_t)((ci & 1) ? (xw[ci >> 1] >> 16) : (xw[ci >> 1] & 0xFFFFu));
                    *(bf16_t*)(xbase + px * C1_ROWX + ((((off >> 4) ^ c1_swz_x(px)) << 4) | (off & 15))) = v;
                }
            }
        }
    };

    // wave w: T rows of d channels 16 w .. 16 w + 15 (acc[.][0]) and of y channels 16 w .. (acc[.][1]); accd = d^T y of
    // those two blocks, whose diagonal is sum_p d * y; wave 0 also the all-ones row
    f32x4 acc[2][2], acc1[2], accd = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // all-ones operand: channel (lane & 15) == 0 of its block is 1.0 for every k, the rest 0
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
    const unsigned short one = (lane & 15) == 0 ? (unsigned short)0x3F80 : (unsigned short)0;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (u16x8){one, one, one, one, one, one, one, one});

    // the matrix work of the tile in stage `buf`: one half of K at a time (fewer live fragments)
    auto compute = [&](int buf) {
        const unsigned abase = (unsigned)(size_t)(lptr_t)smem + (unsigned)(buf * C1_STAGE), xbase = abase + C1_A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 xa[2], cb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xa[i] = c1_frag<true>(xbase, kk * 32, i * 16, lane);
                cb[i] = c1_frag<false>(abase, kk * 32, i * 64 + wave * 16, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) { c1_wait(xa[i]); c1_wait(cb[i]); }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i], cb[j], acc[i][j], 0, 0, 0);
                if (wave == 0) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[i], ones, acc1[i], 0, 0, 0);
            }
            if (MODE != 1) accd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[0], cb[1], accd, 0, 0, 0);
        }
    };

    // MODE 2: the y half of stage `buf` from its (complete) X tile: wave w makes pixels 16 w .. 16 w + 15
    bf16x8 wfrag[4];
    if (MODE == 2) {
        __syncthreads();                   // wlds, bias_l
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) wfrag[cb] = *(const bf16x8*)(wlds + (16 * cb + (lane & 15)) * 64 + (lane >> 4) * 16);
    }
    auto make_y = [&](int buf, int tile) {
        unsigned char* abase = smem + buf * C1_STAGE;
        const unsigned char* xbase = abase + C1_A_BYTES;
        const int row = 16 * wave + (lane & 15), q = lane >> 4;
        const bf16x8 xb = *(const bf16x8*)(xbase + row * C1_ROWX + ((q ^ c1_swz_x(row)) << 4));
        const bool valid = tile * C1_BK + row < p.npix;
        const int sa = c1_swz_a(row);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            // result register r: channel 16 cb + 4 q + r of pixel `row`
            const f32x4 yv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[cb], xb, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const float4 b4 = *(const float4*)(bias_l + 16 * cb + 4 * q);
            uint2 o = make_uint2(pack2bf(yv[0] + b4.x, yv[1] + b4.y), pack2bf(yv[2] + b4.z, yv[3] + b4.w));
            if (!valid) o = make_uint2(0u, 0u);
            *(uint2*)(abase + row * C1_ROWA + (((8 + 2 * cb + (q >> 1)) ^ sa) << 4) + (q & 1) * 8) = o;
        }
    };

    // A tile's registers are free as soon as it is staged: the loads of the tile after next go out right then and have
    // the barrier and the next tile's matrix work to land.
    const int G = gridDim.x;
    int tile = blockIdx.x;
    fetch(r0, tile);
    __syncthreads();                       // the clear and the ones column are complete
    deposit(r0, 0, tile);
    if (tile + G < p.ntile) fetch(r0, tile + G);
    if (MODE == 2) {
        barrier_keep_copies();             // the X tile is complete
        make_y(0, tile);
    }
    barrier_keep_copies();
    int buf = 0;
    while (true) {
        const bool more = tile + G < p.ntile;
        if (MODE == 2) {
            // the next tile's d and X first, alone; then its y (which needs the complete X tile) beside this tile's
            // matrix work: the two have no data in common, so waves in either overlap
            if (more) {
                deposit(r0, buf ^ 1, tile + G);
                if (tile + 2 * G < p.ntile) fetch(r0, tile + 2 * G);
                barrier_keep_copies();
                make_y(buf ^ 1, tile + G);
            }
            compute(buf);
        } else {
            compute(buf);
            if (more) {
                deposit(r0, buf ^ 1, tile + G);
                if (tile + 2 * G < p.ntile) fetch(r0, tile + 2 * G);
            }
        }
        barrier_keep_copies();             // stage buf is read out, stage buf ^ 1 is written
        if (!more) break;
        tile += G;
        buf ^= 1;
    }

    // acc[i][j][r]: column 16 i + 4 (lane >> 4) + r of T row 64 j + 16 wave + (lane & 15)
    float* out = p.part + (long)blockIdx.x * C1_ROWF;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = 16 * i + 4 * (lane >> 4);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *(float4*)(out + (64 * j + 16 * wave + (lane & 15)) * C1_COLS + col) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (wave == 0 && (lane & 15) == 0)
            *(float4*)(out + 128 * C1_COLS + col) = make_float4(acc1[i][0], acc1[i][1], acc1[i][2], acc1[i][3]);
    }
    // accd[r]: d channel 4 (lane >> 4) + r against y channel (lane & 15) of the wave's block: keep the diagonal
    const int r = (lane & 15) - 4 * (lane >> 4);
    if (MODE != 1 && r >= 0 && r < 4)
        out[C1_TROWS * C1_COLS + 16 * wave + (lane & 15)] = r == 0 ? accd[0] : r == 1 ? accd[1] : r == 2 ? accd[2] : accd[3];
}

// The tail of both forms in ONE launch: workgroup c folds, over the nb partial rows, the T rows channel c needs (c,
// 64 + c, the ones row and -- first conv -- its sum d * y) in fp64 and fixed order, then forms channel c's results.
//   MODE 0: dW[c][Cin][9] and sums = [d gamma (64) | d beta (64)];  out2 = dW
//   MODE 1: sums = [sum g * xhat (64) | sum g (64)], out2 = sums_head = [dW (CO x 64) | db (CO)]
// block = 1024 = 32 columns x 32 row groups; a launch like this is a chain of memory round trips, so eight rows are
// requested at a time.
template <int MODE>
__global__ __launch_bounds__(1024) void c1_fold_kernel(const float* __restrict__ part, int nb, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ w,
                                                       double inv_count, float* __restrict__ sums, float* __restrict__ out2,
                                                       int NC) {
    constexpr int NQ = MODE == 0 ? 4 : 3;
    __shared__ double red[NQ][32][33];
    __shared__ double T[4][32];
    const int c = blockIdx.x, col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    // the NQ rows are folded side by side: all their loads of a batch of eight partial rows are in flight together
    long off[NQ];
    double acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        off[q] = (q == 0 ? c * C1_COLS : q == 1 ? (64 + c) * C1_COLS : q == 2 ? 128 * C1_COLS : C1_TROWS * C1_COLS + c) +
                 (q < 3 ? col : 0);            // the fourth row is one value: every lane reads it, lane 0's copy is used
        acc[q] = 0.0;
    }
    int r = rg;
    for (; r + 224 < nb; r += 256) {
        float v[NQ][8];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[q][i] = part[(long)(r + 32 * i) * C1_ROWF + off[q]];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            acc[q] += (((double)v[q][0] + (double)v[q][1]) + ((double)v[q][2] + (double)v[q][3])) +
                      (((double)v[q][4] + (double)v[q][5]) + ((double)v[q][6] + (double)v[q][7]));
    }
    for (; r < nb; r += 32)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] += (double)part[(long)r * C1_ROWF + off[q]];
#pragma unroll
    for (int q = 0; q < NQ; ++q) red[q][rg][col] = acc[q];
    __syncthreads();
    if (rg < NQ) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[rg][i][col];
        T[rg][col] = t;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (MODE == 0) {
        const int Cin = NC;
        const double s_d = T[0][31];
        const double s_dx = (double)rstd[c] * (T[3][0] - (double)mean[c] * s_d);      // sum d * xhat
        if (t < 9 * Cin) {
            const int ci = t / 9, tap = t - ci * 9, j = tap * Cin + ci;
            const double t3 = (double)rstd[c] * (T[1][j] - (double)mean[c] * T[2][j]);
            out2[c * 9 * Cin + t] = (float)((double)scale[c] * (T[0][j] - s_d * inv_count * T[2][j] - s_dx * inv_count * t3));
        }
        if (t == 0) { sums[c] = (float)s_dx; sums[64 + c] = (float)s_d; }
    } else {
        const int CO = NC;
        if (t == 0) {
            double s_g = 0.0, s_gy = 0.0;
            for (int o = 0; o < CO; ++o) {
                const double p2 = T[0][o] + T[0][4 + o], p1 = T[1][o] + T[1][4 + o];
                s_g += (double)w[o * 64 + c] * p1;
                s_gy += (double)w[o * 64 + c] * p2;
                out2[o * 64 + c] = (float)((double)scale[c] * p2 + (double)shift[c] * p1);
            }
            sums[c] = (float)((double)rstd[c] * (s_gy - (double)mean[c] * s_g));
            sums[64 + c] = (float)s_g;
            if (c < CO) out2[CO * 64 + c] = (float)(T[2][c] + T[2][4 + c]);
        }
    }
}

static int c1_blocks(long npix) {
    const long nt = (npix + C1_BK - 1) / C1_BK;
    return (int)(nt < C1_MAX_BLOCKS ? nt : C1_MAX_BLOCKS);
}
static int c1_shift(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

long head_sums_scratch_floats(long npix) { return ((long)c1_blocks(npix) + 1) * C1_ROWF; }

// the sums of vae_head_bwd (norm.hip) on the matrix pipe; scratch: head_sums_scratch_floats(npix) floats
int head_sums_mfma(const float* drecon, const float* recon, const bf16_t* y, const float* scale, const float* shift,
                   const float* mean, const float* rstd, const float* w, float* scratch, float* sums, float* sums_head,
                   long npix, int hw, int CO, float* dl_out, hipStream_t s) {
    Conv1Bwd p = {};
    p.dl_out = dl_out;
    p.y = y; p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd; p.part = scratch;
    p.npix = (int)npix; p.ntile = (int)((npix + C1_BK - 1) / C1_BK);
    p.drecon = drecon; p.recon = recon; p.hw = hw; p.hwshift = c1_shift(hw);
    const int nb = c1_blocks(npix);
    switch (CO) {
        case 1: hipLaunchKernelGGL((bn_bwd_conv1_kernel<1, 1>), dim3(nb), dim3(256), 0, s, p); break;
        case 2: hipLaunchKernelGGL((bn_bwd_conv1_kernel<1, 2>), dim3(nb), dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL((bn_bwd_conv1_kernel<1, 3>), dim3(nb), dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((bn_bwd_conv1_kernel<1, 4>), dim3(nb), dim3(256), 0, s, p); break;
    }
    VAE_LAUNCH_CHECK("bn_bwd_conv1_kernel<head>");
    hipLaunchKernelGGL(c1_fold_kernel<1>, dim3(64), dim3(1024), 0, s, scratch, nb, scale, shift, mean, rstd, w, 0.0, sums, sums_head, CO);
    VAE_LAUNCH_CHECK("c1_fold_kernel<head>");
    return 0;
}

}  // namespace vae

using namespace vae;

extern "C" long vae_bn_bwd_conv1_scratch_floats(int n_img, int H, int W) {
    if (n_img <= 0 || H <= 0 || W <= 0) return -1;
    return ((long)c1_blocks((long)n_img * H * W) + 1) * C1_ROWF;
}

extern "C" int vae_bn_bwd_conv1(const void* gin, const void* y, const void* x8, const float* scale, const float* shift,
                                const float* mean, const float* rstd, float* scratch, float* sums, float* dW, int n_img,
                                int H, int W, int Cin, void* stream) {
    VAE_CHECK(gin && y && x8 && scale && shift && mean && rstd && scratch && sums && dW, "vae_bn_bwd_conv1: null pointer");
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31) - 64 * C1_MAX_BLOCKS,
              "vae_bn_bwd_conv1: bad geometry n=%d H=%d W=%d", n_img, H, W);
    VAE_CHECK(Cin >= 1 && Cin <= 3, "vae_bn_bwd_conv1: Cin=%d out of range (1..3: nine taps of Cin channels must fit 32 columns)", Cin);
    VAE_CHECK((((uintptr_t)gin | (uintptr_t)y | (uintptr_t)scratch) & 15) == 0 && ((uintptr_t)x8 & 7) == 0,
              "vae_bn_bwd_conv1: gin, y and scratch must be 16-byte aligned, x8 8-byte aligned");
    const long npix = (long)n_img * H * W;
    Conv1Bwd p = {};
    p.gin = (const bf16_t*)gin; p.y = (const bf16_t*)y; p.x8 = (const bf16_t*)x8;
    p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd; p.part = scratch;
    p.npix = (int)npix; p.H = H; p.W = W; p.wshift = c1_shift(W); p.hshift = c1_shift(H); p.Cin = Cin;
    p.ntile = (int)((npix + C1_BK - 1) / C1_BK);
    const int nb = c1_blocks(npix);
    hipStream_t s = (hipStream_t)stream;
    {
        // gin and y read once, the 8-channel input once
        ProfScope prof("bn_bwd_conv1_kernel", 0.0, (double)npix * (2.0 * 64 * 2 + 16), s);
        if (Cin == 1) hipLaunchKernelGGL((bn_bwd_conv1_kernel<0, 1>), dim3(nb), dim3(256), 0, s, p);
        else if (Cin == 2) hipLaunchKernelGGL((bn_bwd_conv1_kernel<0, 2>), dim3(nb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((bn_bwd_conv1_kernel<0, 3>), dim3(nb), dim3(256), 0, s, p);
        VAE_LAUNCH_CHECK("bn_bwd_conv1_kernel");
    }
    hipLaunchKernelGGL(c1_fold_kernel<0>, dim3(64), dim3(1024), 0, s, scratch, nb, scale, shift, mean, rstd, (const float*)nullptr,
                       1.0 / (double)npix, sums, dW, Cin);
    VAE_LAUNCH_CHECK("c1_fold_kernel");
    return 0;
}

// The same without the stored conv output: `relu_bits` (one bit per element: the forward's ReLU decision, as
// vae_conv3x3_c8_bn_act leaves it) in place of y, and the conv's own weight / bias, from which each tile's y is rebuilt.
extern "C" int vae_bn_bwd_conv1_bits(const void* gin, const void* relu_bits, const void* x8, const float* w, const float* bias,
                                     const float* scale, const float* mean, const float* rstd, float* scratch, float* sums,
                                     float* dW, int n_img, int H, int W, int Cin, void* stream) {
    VAE_CHECK(gin && relu_bits && x8 && w && scale && mean && rstd && scratch && sums && dW, "vae_bn_bwd_conv1_bits: null pointer");
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31) - 64 * C1_MAX_BLOCKS,
              "vae_bn_bwd_conv1_bits: bad geometry n=%d H=%d W=%d", n_img, H, W);
    VAE_CHECK(Cin >= 1 && Cin <= 3, "vae_bn_bwd_conv1_bits: Cin=%d out of range (1..3: nine taps of Cin channels must fit 32 columns)", Cin);
    VAE_CHECK((((uintptr_t)gin | (uintptr_t)scratch) & 15) == 0 && ((uintptr_t)x8 & 7) == 0,
              "vae_bn_bwd_conv1_bits: gin and scratch must be 16-byte aligned, x8 8-byte aligned");
    const long npix = (long)n_img * H * W;
    Conv1Bwd p = {};
    p.gin = (const bf16_t*)gin; p.relu_bits = (const unsigned char*)relu_bits; p.x8 = (const bf16_t*)x8;
    p.w = w; p.bias = bias; p.part = scratch;
    p.npix = (int)npix; p.H = H; p.W = W; p.wshift = c1_shift(W); p.hshift = c1_shift(H); p.Cin = Cin;
    p.ntile = (int)((npix + C1_BK - 1) / C1_BK);
    const int nb = c1_blocks(npix);
    hipStream_t s = (hipStream_t)stream;
    {
        // gin read once, a byte of bits per 8 elements, the 8-channel input once
        ProfScope prof("bn_bwd_conv1_kernel", 0.0, (double)npix * (64 * 2 + 8 + 16), s);
        if (Cin == 1) hipLaunchKernelGGL((bn_bwd_conv1_kernel<2, 1>), dim3(nb), dim3(256), 0, s, p);
        else if (Cin == 2) hipLaunchKernelGGL((bn_bwd_conv1_kernel<2, 2>), dim3(nb), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((bn_bwd_conv1_kernel<2, 3>), dim3(nb), dim3(256), 0, s, p);
        VAE_LAUNCH_CHECK("bn_bwd_conv1_kernel<bits>");
    }
    hipLaunchKernelGGL(c1_fold_kernel<0>, dim3(64), dim3(1024), 0, s, scratch, nb, scale, (const float*)nullptr, mean, rstd,
                       (const float*)nullptr, 1.0 / (double)npix, sums, dW, Cin);
    VAE_LAUNCH_CHECK("c1_fold_kernel");
    return 0;
}
