// Backward of a 64 -> 64 ConvTranspose2d(kernel 2, stride 2) in one kernel (vae/models/vae.py:131, the last upsampling
// of the decoder): input gradient AND weight gradient from ONE read of the output gradient.
//
//   dX[i][ci]          = sum_{tap, co} dY[out(i, tap)][co] * W[ci][co][tap]          (i: input pixel, out(): its 2x2 block)
//   dW[ci][co][tap]    = sum_i        X[i][ci]            * dY[out(i, tap)][co]
//
// With stride = kernel the four output pixels of an input pixel belong to it alone, so a tile of 64 input pixels needs
// exactly 256 rows of dY (64 pixels x 4 taps) and both products contract over what the tile holds: per pixel the 256
// values G[i][k] (k = tap * 64 + co) are one LDS row.  dX = G * W^T contracts over k (plain 16-byte fragment reads of
// the row against the weight matrix resident in LDS), dW = X^T * G contracts over the pixels (transposing reads of the
// same rows and of the X tile).  As two kernels the weight gradient read dY once and X four times (one tap per
// workgroup) and the input gradient read dY again: 200 MB instead of 100 at batch 512.
// Tiles are copied global -> LDS (16 bytes per lane, no registers), three stages, a counted wait per tile.  A workgroup
// keeps dW in registers over all its tiles and leaves one split-K slab [co][tap][ci], folded with the others of the
// backward stage.
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

int g_convt_bwd_mode = 0;

constexpr int CB_BK = 64;                                    // input pixels per tile
constexpr int CB_ROWP = 512, CB_ROWQ = 128;                  // LDS row bytes: 4 taps x 64 co | 64 ci
constexpr int CB_P = CB_BK * CB_ROWP, CB_Q = CB_BK * CB_ROWQ, CB_STAGE = CB_P + CB_Q, CB_W = 64 * CB_ROWP;
constexpr int CB_NW = 8;                                     // waves: two per SIMD cover each other's copy issue and LDS latency
constexpr int CB_PI = CB_P / 1024 / CB_NW, CB_QI = CB_Q / 1024 / CB_NW, CB_WI = CB_W / 1024 / CB_NW;   // copy instructions per wave
constexpr int CB_NST = 3;                                    // tiles in LDS: one being worked on, two arriving
constexpr int CB_MAX_BLOCKS = 256;                           // one workgroup per CU (152 KB of LDS)

struct ConvtBwd {
    const bf16_t* g;      // dY [n][2H][2W][64]
    const bf16_t* a;      // X  [n][H][W][64]
    const bf16_t* wd;     // W  [ci][tap * 64 + co] (the packed operand of the input-gradient GEMM)
    bf16_t* dx;           // [n][H][W][64]
    float* slab;          // [workgroup][co][tap][ci]
    int npix, H, W, wshift, hshift, ntile;
};

typedef const __attribute__((address_space(1))) void* cb_gptr_t;
typedef __attribute__((address_space(3))) void* cb_lptr_t;

// chunk (16 bytes) swizzles: rows 512 bytes apart (the dY tile, the weights) and 128 bytes apart (the X tile); the 8 rows
// {b..b+3, b+8..b+11} a half of a transposing read touches land on 8 distinct 32-byte slots of the 256-byte bank line
__device__ __forceinline__ int cb_swz_p(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }
__device__ __forceinline__ int cb_swz_q(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }
__device__ __forceinline__ int cb_swz_w(int row) { return row & 15; }

// transposing fragment: lanes 16g..16g+15 fetch rows k0+8g .. k0+8g+7 of columns col0..col0+15; lane (l & 15) gets
// column col0 + (l & 15), 8 consecutive rows.  Inline asm (the compiler would drain the copies in flight before a
// transposing read of its own): cb_wait before the first use.
template <int ROWB>
__device__ __forceinline__ bf16x8 cb_tr_frag(unsigned base, int k0, int col0, int lane) {
    const int g = lane >> 4, t = lane & 15;
    const int row = k0 + 8 * g + (t >> 2);
    const int off = (col0 + 4 * (t & 3)) * 2;
    const int chunk = off >> 4, within = off & 15;
    const int s0 = ROWB == CB_ROWP ? cb_swz_p(row) : cb_swz_q(row), s1 = ROWB == CB_ROWP ? cb_swz_p(row + 4) : cb_swz_q(row + 4);
    const unsigned a = base + row * ROWB + (((chunk ^ s0) << 4) | within);
    const unsigned b = base + (row + 4) * ROWB + (((chunk ^ s1) << 4) | within);
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b) : "memory");
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void cb_wait(bf16x8& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f) : : "memory"); }

__global__ __launch_bounds__(64 * CB_NW, 1) void convt_bwd_fused_kernel(const ConvtBwd p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[CB_NST * CB_STAGE + CB_W];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wq = wave & 3, wh = wave >> 2;      // pixel block / tap, and half of the input channels
    unsigned char* const wl = smem + CB_NST * CB_STAGE;
    const bf16_t* zero = (const bf16_t*)g_zero_page;

    // weights: LDS[ci][cl] = W[ci][cl ^ swz(ci)], two rows of 32 chunks per copy instruction
#pragma unroll
    for (int i = 0; i < CB_WI; ++i) {
        const int t = wave + CB_NW * i;
        const int row = 2 * t + (lane >> 5), c = (lane & 31) ^ cb_swz_w(row);
        __builtin_amdgcn_global_load_lds((cb_gptr_t)(p.wd + row * 256 + c * 8), (cb_lptr_t)(wl + t * 1024), 16, 0, 0);
    }
    // one tile: CB_PI + CB_QI copy instructions per wave
    auto stage = [&](int buf, int tile) {
        unsigned char* pbase = smem + buf * CB_STAGE;
        unsigned char* qbase = pbase + CB_P;
#pragma unroll
        for (int i = 0; i < CB_PI; ++i) {
            const int t = wave + CB_NW * i;
            const int row = 2 * t + (lane >> 5), c = (lane & 31) ^ cb_swz_p(row);
            const int k = tile * CB_BK + row, tap = c >> 3;
            const bf16_t* src = zero;
            if (k < p.npix) {
                int h, w, img;
                if (p.wshift >= 0 && p.hshift >= 0) { w = k & (p.W - 1); h = (k >> p.wshift) & (p.H - 1); img = k >> (p.wshift + p.hshift); }
                else { w = k % p.W; const int q = k / p.W; h = q % p.H; img = q / p.H; }
                const long r = ((long)img * 2 * p.H + 2 * h + (tap >> 1)) * (2 * p.W) + 2 * w + (tap & 1);
                src = p.g + r * 64 + (c & 7) * 8;
            }
            __builtin_amdgcn_global_load_lds((cb_gptr_t)src, (cb_lptr_t)(pbase + t * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < CB_QI; ++i) {
            const int t = wave + CB_NW * i;
            const int row = 8 * t + (lane >> 3), c = (lane & 7) ^ cb_swz_q(row);
            const int k = tile * CB_BK + row;
            const bf16_t* src = k < p.npix ? p.a + (long)k * 64 + c * 8 : zero;
            __builtin_amdgcn_global_load_lds((cb_gptr_t)src, (cb_lptr_t)(qbase + t * 1024), 16, 0, 0);
        }
    };

    // wave (wq, wh): weight gradient of tap wq (k = 64 wq .. 64 wq + 63) x input channels 32 wh .. 32 wh + 31; input
    // gradient of pixels 16 wq .. 16 wq + 15 for the same channels
    f32x4 accw[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) accw[kb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int G = gridDim.x;
    int tile = blockIdx.x;
    stage(0, tile);
    if (tile + G < p.ntile) stage(1, tile + G);
    if (tile + 2 * G < p.ntile) stage(2, tile + 2 * G);
    int buf = 0, it = 0;
    constexpr int SC = CB_PI + CB_QI, ST = 2;      // per tile and wave: copies, input-gradient stores
    for (; tile < p.ntile; tile += G, ++it) {
        // The tile's copies have landed when no more operations are in flight than were issued after them: the copies of
        // the next two tiles and the stores of the last two (whichever exist; the host guarantees whole tiles, so the
        // stores are never skipped).  Issue order: W S0 S1 S2 | st0 S3 | st1 S4 | st2 S5 | ...
        const int ahead = tile + 2 * G < p.ntile ? 2 : tile + G < p.ntile ? 1 : 0;
        if (ahead == 2) {
            if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SC) : "memory");
            else if (it == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SC + ST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * SC + 2 * ST) : "memory");
        } else if (ahead == 1) {
            if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SC) : "memory");
            else if (it == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SC + ST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SC + 2 * ST) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        barrier_keep_copies();
        const unsigned char* pb = smem + buf * CB_STAGE;
        const unsigned pbase = (unsigned)(size_t)(cb_lptr_t)smem + (unsigned)(buf * CB_STAGE), qbase = pbase + CB_P;

        // ---- input gradient: D[ci (rows)][pixel (columns)] over k = 256
        f32x4 accd[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) accd[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int rowd = 16 * wq + l15;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int c = ks * 4 + g4;
            const bf16x8 gf = *(const bf16x8*)(pb + rowd * CB_ROWP + ((c ^ cb_swz_p(rowd)) << 4));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int ci = (2 * wh + nb) * 16 + l15;
                const bf16x8 wf = *(const bf16x8*)(wl + ci * CB_ROWP + ((c ^ cb_swz_w(ci)) << 4));
                accd[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, gf, accd[nb], 0, 0, 0);
            }
        }
        {
            bf16_t* dst = p.dx + ((long)tile * CB_BK + rowd) * 64 + 32 * wh + 4 * g4;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                *(uint2*)(dst + nb * 16) = make_uint2(pack2bf(accd[nb][0], accd[nb][1]), pack2bf(accd[nb][2], accd[nb][3]));
        }
        // ---- weight gradient: D[ci (rows)][k (columns)] over the 64 pixels
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            bf16x8 gt[4], at[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) gt[j] = cb_tr_frag<CB_ROWP>(pbase, 32 * ps, wq * 64 + j * 16, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) at[j] = cb_tr_frag<CB_ROWQ>(qbase, 32 * ps, (2 * wh + j) * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) cb_wait(gt[j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) cb_wait(at[j]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) accw[kb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[cb], gt[kb], accw[kb][cb], 0, 0, 0);
        }
        barrier_keep_copies();             // every wave is done with this stage
        if (tile + CB_NST * G < p.ntile) stage(buf, tile + CB_NST * G);
        buf = buf + 1 == CB_NST ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // accw[kb][cb][r]: ci = 32 wh + 16 cb + 4 (lane >> 4) + r, co = 16 kb + (lane & 15), tap = wq -> slab [co][tap][ci]
    float* out = p.slab + (long)blockIdx.x * (4 * 64 * 64);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            *(float4*)(out + ((kb * 16 + l15) * 4 + wq) * 64 + 32 * wh + cb * 16 + 4 * g4) =
                make_float4(accw[kb][cb][0], accw[kb][cb][1], accw[kb][cb][2], accw[kb][cb][3]);
}

// The same backward with dY not read but made on the way: dY here is the gradient behind the last BatchNorm + ReLU + 1x1
// head (vae/models/vae.py:132-134), g = scale * (pos * sum_o w[o][c] dl[o] - mean(g) - xhat * mean(g * xhat)), a function of
// the ConvTranspose's own output y, of the logit gradients dl (3-4 floats per pixel) and of per-channel constants.  The
// tile's 256 rows of g are computed from y into LDS instead of copied, so the 67 MB tensor is neither written nor read: y and dl in, dX and the dW slabs out.  Register-staged
// tiles (the next tile's y is in flight during the matrix work), two LDS stages, eight waves.
struct HeadConvtBwd {
    const bf16_t* y;      // [n][2H][2W][64], the ConvTranspose output
    const float* dl;      // [n * 2H * 2W][4] logit gradients
    const float* scale;
    const float* shift;
    const float* mean;
    const float* rstd;
    const float* sums;    // [sum g * xhat (64) | sum g (64)]
    const float* w;       // head weights [CO][64]
    float inv_count;
    const bf16_t* a;
    const bf16_t* wd;
    bf16_t* dx;
    float* slab;
    int npix, H, W, wshift, hshift, ntile;
};

template <int CO>
__global__ __launch_bounds__(64 * CB_NW, 1) void head_convt_bwd_kernel(const HeadConvtBwd p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * CB_STAGE + CB_W];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;
    const int wq = wave & 3, wh = wave >> 2;
    const int px = tid >> 3, cg = tid & 7, c0 = cg * 8;      // staging: this lane's pixel of the tile and its 8 channels
    unsigned char* const wl = smem + 2 * CB_STAGE;
#pragma unroll
    for (int i = 0; i < CB_W / 16 / (64 * CB_NW); ++i) {
        const int idx = tid + 64 * CB_NW * i, row = idx >> 5, cl = idx & 31;
        *(uint4*)(wl + row * CB_ROWP + cl * 16) = *(const uint4*)(p.wd + row * 256 + (cl ^ cb_swz_w(row)) * 8);
    }
    // g = scale * (pos * da - c1 - xhat * c2) with da = sum_o w[o] dl[o], xhat = (y - mean) * rstd, as one affine form in
    // y plus the gated term:  g = pos * sum_o (scale w[o]) dl[o] + (ka * y + kb),  ka = -scale c2 rstd,
    // kb = scale (c2 rstd mean - c1): ten vector operations per element instead of seventeen (the staging of a tile is
    // bound by them).  The ReLU predicate keeps the forward's own two-rounding form.
    float sc[8], sh[8], ka[8], kb[8], swv[CO][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        sc[i] = p.scale[c0 + i]; sh[i] = p.shift[c0 + i];
        const float mu = p.mean[c0 + i], rs = p.rstd[c0 + i];
        const float c2 = p.sums[c0 + i] * p.inv_count, c1 = p.sums[64 + c0 + i] * p.inv_count;
        ka[i] = -sc[i] * c2 * rs;
        kb[i] = sc[i] * (c2 * rs * mu - c1);
#pragma unroll
        for (int o = 0; o < CO; ++o) swv[o][i] = sc[i] * p.w[o * 64 + c0 + i];
    }
    // What one lane has in flight for a tile: its 8 channels of the pixel's four output rows, the pixel of X, and ONE of
    // the four rows' logit gradients (lane cg holds row cg & 3; the eight lanes of a pixel hand them round by
    // ds_bpermute when the row is formed) -- 24 registers, so that two tiles fit beside the accumulators.
    struct InFlight {
        uint4 y[4];
        float4 d;
        uint4 x;
    };
    InFlight r0, r1;          // tiles i + 1 and i + 2 while tile i is in the matrix pipe
    // Buffer loads: a tile past the end is fetched at an out-of-range offset (zeros, no traffic) instead of under a
    // branch, so every pass of the loop issues the same loads and the counted waits the compiler derives let the
    // younger set stay in flight (after a conditional fetch it has to assume the loads were not issued).
    const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (int)((unsigned)p.npix * 512u), 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.dl, 0, (int)((unsigned)p.npix * 64u), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)((unsigned)p.npix * 128u), 0x00020000);
    auto ld16 = [](const __amdgpu_buffer_rsrc_t& rs, unsigned off) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
    };
    auto fetch = [&](InFlight& r, int tile) {
        const bool ok = tile < p.ntile;
        const int k = tile * CB_BK + px;          // whole tiles: k < npix when the tile exists
        int h, w, img;
        if (p.wshift >= 0 && p.hshift >= 0) { w = k & (p.W - 1); h = (k >> p.wshift) & (p.H - 1); img = k >> (p.wshift + p.hshift); }
        else { w = k % p.W; const int q = k / p.W; h = q % p.H; img = q / p.H; }
        const unsigned r00 = (unsigned)(img * 2 * p.H + 2 * h) * (unsigned)(2 * p.W) + (unsigned)(2 * w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned rr = r00 + (unsigned)((j >> 1) * 2 * p.W + (j & 1));
            r.y[j] = ld16(y_rsrc, ok ? rr * 128u + (unsigned)c0 * 2u : 0xFFFFFFFFu);
        }
        {
            const int j = cg & 3;
            const unsigned rr = r00 + (unsigned)((j >> 1) * 2 * p.W + (j & 1));
            r.d = __builtin_bit_cast(float4, ld16(d_rsrc, ok ? rr * 16u : 0xFFFFFFFFu));
        }
        r.x = ld16(x_rsrc, ok ? (unsigned)k * 128u + (unsigned)c0 * 2u : 0xFFFFFFFFu);
    };
    const int lane8 = (lane & ~7) * 4;            // byte index (for ds_bpermute) of the first lane of this pixel
    auto deposit = [&](const InFlight& r, int buf) {
        unsigned char* pbase = smem + buf * CB_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t yw[4] = {r.y[j].x, r.y[j].y, r.y[j].z, r.y[j].w};
            const float dsrc[4] = {r.d.x, r.d.y, r.d.z, r.d.w};
            float dl[CO];
#pragma unroll
            for (int o = 0; o < CO; ++o)
                dl[o] = __int_as_float(__builtin_amdgcn_ds_bpermute(lane8 + 4 * j, __float_as_int(dsrc[o])));
            float o8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = __uint_as_float((i & 1) ? (yw[i >> 1] & 0xFFFF0000u) : (yw[i >> 1] << 16));
                const bool pos = v * sc[i] + sh[i] > 0.f;
                float da = dl[0] * swv[0][i];
#pragma unroll
                for (int o = 1; o < CO; ++o) da = fmaf(dl[o], swv[o][i], da);
                o8[i] = (pos ? da : 0.f) + fmaf(ka[i], v, kb[i]);
            }
            *(uint4*)(pbase + px * CB_ROWP + (((j * 8 + cg) ^ cb_swz_p(px)) << 4)) =
                make_uint4(pack2bf(o8[0], o8[1]), pack2bf(o8[2], o8[3]), pack2bf(o8[4], o8[5]), pack2bf(o8[6], o8[7]));
        }
        *(uint4*)(pbase + CB_P + px * CB_ROWQ + ((cg ^ cb_swz_q(px)) << 4)) = r.x;
    };

    f32x4 accw[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) accw[kb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int G = gridDim.x;
    int tile = blockIdx.x;
    fetch(r0, tile);
    deposit(r0, 0);
    fetch(r0, tile + G);
    fetch(r1, tile + 2 * G);
    __syncthreads();                       // the weights and the first tile are in LDS (the prefetches are waited for here too, once)
    int buf = 0;
    // One tile; rs holds the next tile's inputs and is refilled with those of the tile three ahead.  No branch but the
    // guard of the dX store: a workgroup runs an even number of passes, and a pass past its last tile works on a stage
    // of zeros for X (nothing is added to dW) and stores nothing -- the loads, the stage writes and the barrier are the
    // same on every pass, which is what keeps the compiler's counted waits at two tiles of distance.
    auto one_tile = [&](InFlight& rs) {
        const unsigned char* pb = smem + buf * CB_STAGE;
        const unsigned pbase = (unsigned)(size_t)(cb_lptr_t)smem + (unsigned)(buf * CB_STAGE), qbase = pbase + CB_P;
        f32x4 accd[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) accd[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int rowd = 16 * wq + l15;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int c = ks * 4 + g4;
            const bf16x8 gf = *(const bf16x8*)(pb + rowd * CB_ROWP + ((c ^ cb_swz_p(rowd)) << 4));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int ci = (2 * wh + nb) * 16 + l15;
                const bf16x8 wf = *(const bf16x8*)(wl + ci * CB_ROWP + ((c ^ cb_swz_w(ci)) << 4));
                accd[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, gf, accd[nb], 0, 0, 0);
            }
        }
        if (tile < p.ntile) {
            bf16_t* dst = p.dx + ((long)tile * CB_BK + rowd) * 64 + 32 * wh + 4 * g4;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                *(uint2*)(dst + nb * 16) = make_uint2(pack2bf(accd[nb][0], accd[nb][1]), pack2bf(accd[nb][2], accd[nb][3]));
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            bf16x8 gt[4], at[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) gt[j] = cb_tr_frag<CB_ROWP>(pbase, 32 * ps, wq * 64 + j * 16, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) at[j] = cb_tr_frag<CB_ROWQ>(qbase, 32 * ps, (2 * wh + j) * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) cb_wait(gt[j]);
#pragma unroll
            for (int j = 0; j < 2; ++j) cb_wait(at[j]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) accw[kb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at[cb], gt[kb], accw[kb][cb], 0, 0, 0);
        }
        deposit(rs, buf ^ 1);
        fetch(rs, tile + 3 * G);
        barrier_keep_copies();             // stage buf is read out, stage buf ^ 1 is written
        tile += G;
        buf ^= 1;
    };
    const int mine = (p.ntile - (int)blockIdx.x + G - 1) / G;      // >= 1: the grid is at most ntile wide
    for (int i = 0; i < mine; i += 2) {
        one_tile(r0);
        one_tile(r1);
    }
    float* out = p.slab + (long)blockIdx.x * (4 * 64 * 64);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
            *(float4*)(out + ((kb * 16 + l15) * 4 + wq) * 64 + 32 * wh + cb * 16 + 4 * g4) =
                make_float4(accw[kb][cb][0], accw[kb][cb][1], accw[kb][cb][2], accw[kb][cb][3]);
}

static int cb_shift(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace vae

using namespace vae;

extern "C" int vae_convt2x2_bwd_fused_ok(int n_img, int H, int W, int Cin, int Cout) {
    const long npix = (long)n_img * H * W;
    return g_convt_bwd_mode != 1 && Cin == 64 && Cout == 64 && n_img > 0 && H > 0 && W > 0 && npix % CB_BK == 0 &&
           npix * 4 * 64 * 2 < (1L << 32);
}

extern "C" int vae_convt2x2_bwd_fused_slabs(int n_img, int H, int W) {
    const long nt = ((long)n_img * H * W + CB_BK - 1) / CB_BK;
    return (int)(nt < CB_MAX_BLOCKS ? nt : CB_MAX_BLOCKS);
}

extern "C" int vae_convt2x2_bwd_fused(const void* dy, const void* x, const void* w_packed, void* dx, float* slab, int n_img,
                                      int H, int W, int Cin, int Cout, void* stream) {
    VAE_CHECK(dy && x && w_packed && dx && slab, "vae_convt2x2_bwd_fused: null pointer");
    VAE_CHECK(Cin == 64 && Cout == 64, "vae_convt2x2_bwd_fused: 64 -> 64 channels only (got %d -> %d)", Cin, Cout);
    const long npix = (long)n_img * H * W;
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && npix % CB_BK == 0 && npix * 4 * 64 * 2 < (1L << 32),
              "vae_convt2x2_bwd_fused: n*H*W = %ld must be a positive multiple of %d (and the output below 4 GB)", npix, CB_BK);
    VAE_CHECK((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)dx | (uintptr_t)slab) & 15) == 0,
              "vae_convt2x2_bwd_fused: operands must be 16-byte aligned");
    ConvtBwd p = {};
    p.g = (const bf16_t*)dy; p.a = (const bf16_t*)x; p.wd = (const bf16_t*)w_packed; p.dx = (bf16_t*)dx; p.slab = slab;
    p.npix = (int)npix; p.H = H; p.W = W; p.wshift = cb_shift(W); p.hshift = cb_shift(H); p.ntile = (int)(npix / CB_BK);
    const int nb = vae_convt2x2_bwd_fused_slabs(n_img, H, W);
    hipStream_t s = (hipStream_t)stream;
    // dY and X read once, dX written once, one slab per workgroup
    ProfScope prof("convt_bwd_fused_kernel", 2.0 * 2.0 * npix * 256.0 * 64.0, npix * (4.0 * 128 + 128 + 128) + nb * 65536.0, s);
    hipLaunchKernelGGL(convt_bwd_fused_kernel, dim3(nb), dim3(64 * CB_NW), 0, s, p);
    VAE_LAUNCH_CHECK("convt_bwd_fused_kernel");
    return 0;
}

extern "C" int vae_head_convt_bwd_fused(const void* y, const float* dl, const float* scale, const float* shift, const float* mean,
                                        const float* rstd, const float* sums, const float* w_head, int Cout_head, const void* x,
                                        const void* w_packed, void* dx, float* slab, int n_img, int H, int W, void* stream) {
    VAE_CHECK(y && dl && scale && shift && mean && rstd && sums && w_head && x && w_packed && dx && slab, "vae_head_convt_bwd_fused: null pointer");
    VAE_CHECK(Cout_head >= 1 && Cout_head <= 4, "vae_head_convt_bwd_fused: head out_channels=%d not supported (1..4)", Cout_head);
    const long npix = (long)n_img * H * W;
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && npix % CB_BK == 0 && npix * 4 * 64 * 2 < (1L << 32),
              "vae_head_convt_bwd_fused: n*H*W = %ld must be a positive multiple of %d (and the output below 4 GB)", npix, CB_BK);
    VAE_CHECK((((uintptr_t)y | (uintptr_t)dl | (uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)dx | (uintptr_t)slab) & 15) == 0,
              "vae_head_convt_bwd_fused: operands must be 16-byte aligned");
    HeadConvtBwd p = {};
    p.y = (const bf16_t*)y; p.dl = dl; p.scale = scale; p.shift = shift; p.mean = mean; p.rstd = rstd; p.sums = sums; p.w = w_head;
    p.inv_count = (float)(1.0 / (4.0 * (double)npix));
    p.a = (const bf16_t*)x; p.wd = (const bf16_t*)w_packed; p.dx = (bf16_t*)dx; p.slab = slab;
    p.npix = (int)npix; p.H = H; p.W = W; p.wshift = cb_shift(W); p.hshift = cb_shift(H); p.ntile = (int)(npix / CB_BK);
    const int nb = vae_convt2x2_bwd_fused_slabs(n_img, H, W);
    hipStream_t s = (hipStream_t)stream;
    // y and dl read once, X read once, dX written once, one slab per workgroup
    ProfScope prof("head_convt_bwd_kernel", 2.0 * 2.0 * npix * 256.0 * 64.0, npix * (4.0 * 128 + 4.0 * 16 + 128 + 128) + nb * 65536.0, s);
    switch (Cout_head) {
        case 1: hipLaunchKernelGGL(head_convt_bwd_kernel<1>, dim3(nb), dim3(64 * CB_NW), 0, s, p); break;
        case 2: hipLaunchKernelGGL(head_convt_bwd_kernel<2>, dim3(nb), dim3(64 * CB_NW), 0, s, p); break;
        case 3: hipLaunchKernelGGL(head_convt_bwd_kernel<3>, dim3(nb), dim3(64 * CB_NW), 0, s, p); break;
        default: hipLaunchKernelGGL(head_convt_bwd_kernel<4>, dim3(nb), dim3(64 * CB_NW), 0, s, p); break;
    }
    VAE_LAUNCH_CHECK("head_convt_bwd_kernel");
    return 0;
}
