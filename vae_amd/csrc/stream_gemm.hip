// Streaming implicit GEMM for launches with a short K over many pixels: the first conv (3 input channels padded to 8,
// K = 9 taps x 8) and the ConvTranspose2d k2 s2 forward (K = Cin) of the large decoder layers.
//
//   out[m][n] = sum_k A[m][k] * Wt[n][k]      (bf16 in, fp32 accumulate, bf16 out)
//
// Such a launch moves tens of MB for a few GFLOP: it is a stream, and what a tiled GEMM spends per tile -- staging both
// operands through LDS, a barrier per K-step, a prologue and an epilogue per tile -- is all overhead.  Here
//   * the weights of a workgroup's 64*NCH output columns sit in LDS for the whole launch, already in MFMA fragment
//     order (one conflict-free ds_read_b128 per fragment);
//   * a wave owns 32 pixels at a time and loads their K values straight from global memory into the MFMA "B" operand
//     registers (lane = pixel l & 15, 16-byte K chunk l >> 4): K is contiguous in NHWC memory, so no transposition
//     and no LDS round trip; padding pixels and unused taps are out-of-range buffer offsets, which read as zero;
//   * the loads of the next 32 pixels are issued before the current ones are multiplied; there is no barrier in the
//     loop and waves run independently;
//   * the rows of a weight fragment are a permutation of the output channels chosen so that a lane ends up with two
//     runs of 8 consecutive channels of its pixel: two 16-byte stores per pixel and 64-channel chunk, 64 bytes
//     contiguous across the four lanes of a pixel;
//   * per-channel sums for the BatchNorm that follows stay in registers for the whole launch; one statistics row per
//     workgroup.
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

int g_stream_gemm_mode = 0;    // 0: where it applies and pays, 1: never, 2: wherever it applies (tests, A-B measurements)

enum { SA_PLAIN = 0, SA_C8 = 1 };
// SE_STATS: nothing is stored, only the statistics rows leave (the first of two passes over an input that is cheaper
// to convolve twice than to write out and read back).  SE_BNACT: the second pass -- BatchNorm + ReLU applied to the
// bf16-rounded result, the activation stored instead of the conv output, and (optionally) one bit per element that
// says whether the ReLU passed.
enum { SE_BF16 = 0, SE_CONVT = 1, SE_STATS = 2, SE_BNACT = 3 };

struct StreamGemm {
    const bf16_t* A;
    const bf16_t* Wt;     // [N][ldw], K contiguous
    const float* bias;    // [N] ([Co] for SE_CONVT), may be null
    bf16_t* out;
    float* stat_sum;      // [gridDim.x][N / NCH], may be null
    float* stat_sq;
    int M, N;
    int ldw, lda, ldo;
    int H, W;             // pixel grid of the GEMM rows
    int wshift, hshift;
    int Co;               // SE_CONVT: channels per tap of the output
    unsigned a_bytes;
    int npairs;           // M / 32
    const float* scale;   // SE_BNACT: [N] each
    const float* shift;
    unsigned char* relu_bits;   // SE_BNACT, optional: [M][N / 8], bit (c & 7) of byte c >> 3 of a pixel = "channel c passed the ReLU"
};

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4v;

__device__ __forceinline__ float row16_sum_s(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// channel (0..63) held in row `rho` (0..15) of weight tile j (0..3) of a 64-channel chunk: lane group g = rho >> 2
// collects channels g*8 .. g*8+7 from tiles 0 and 1 and 32 + g*8 .. from tiles 2 and 3
__host__ __device__ __forceinline__ int chunk_channel(int j, int rho) {
    return (j >> 1) * 32 + (rho >> 2) * 8 + (j & 1) * 4 + (rho & 3);
}

// KS: K / 32.  NCH: 64-column chunks per workgroup, worked one after the other on the same pixel registers (only with
// KS <= 4, where all of K is in registers).  K is walked in chunks of KC = min(KS, 4) steps.
template <int AMODE, int EPI, int KS, int NCH>
__global__ __launch_bounds__(256, 3) void stream_gemm_kernel(const StreamGemm p) {
    constexpr int KC = KS < 4 ? KS : 4;
    constexpr int NKC = KS / KC;
    static_assert(KS % KC == 0, "K-steps must be a multiple of the chunk");
    static_assert(NCH == 1 || NKC == 1, "several column chunks need all of K in registers");
    constexpr int W_BYTES = NCH * 4 * KS * 1024;
    __shared__ __attribute__((aligned(16))) unsigned char wsm[W_BYTES];
    __shared__ __attribute__((aligned(16))) float bias_s[NCH * 64];
    __shared__ float red[4][2][64];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, px = lane & 15;
    const int n0 = blockIdx.y * (64 * NCH);

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFFFu;

    // SA_C8: tap of this lane's 16-byte chunk in K-step ks is ks*4 + g
    int c8_d[KC];          // packed: (dh + 1) | (dw + 1) << 2 | valid << 4
    if (AMODE == SA_C8) {
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
            const int t = ks * 4 + g;
            c8_d[ks] = t < 9 ? ((t / 3) | ((t % 3) << 2) | 16) : 0;
        }
    }

    auto load_item = [&](int pair, int kc, u32x4v (&dst)[2][KC]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = pair * 32 + i * 16 + px;
            const int w = m & (p.W - 1), h = (m >> p.wshift) & (p.H - 1);
            unsigned base = 0;
            if (AMODE == SA_PLAIN) base = (unsigned)m * (unsigned)p.lda * 2u;
            if (AMODE == SA_C8) base = (unsigned)m * 16u;
#pragma unroll
            for (int ks = 0; ks < KC; ++ks) {
                unsigned off;
                if (AMODE == SA_PLAIN) {
                    off = base + (unsigned)(((kc * KC + ks) * 4 + g) * 16);
                } else {
                    const int d = c8_d[ks];
                    const int dh = (d & 3) - 1, dw = ((d >> 2) & 3) - 1;
                    const bool ok = (d & 16) && (unsigned)(h + dh) < (unsigned)p.H && (unsigned)(w + dw) < (unsigned)p.W;
                    off = ok ? base + (unsigned)((dh * p.W + dw) * 16) : OOB;
                }
                dst[i][ks] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)off, 0, 0));
            }
        }
    };

    float ssum[16], ssq[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ssum[e] = ssq[e] = 0.f;
    const bool want_stats = EPI != SE_BNACT && p.stat_sum != nullptr;
    // SE_BNACT (one column chunk): scale / shift of this lane's 16 channels, in the order of ssum's slots
    float bsc[16], bsh[16];
    if (EPI == SE_BNACT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int c = n0 + (e >> 3) * 32 + g * 8 + (e & 7);
            bsc[e] = p.scale[c];
            bsh[e] = p.shift[c];
        }
    }

    const int nw = gridDim.x * 4;
    int pair = blockIdx.x * 4 + wave;
    u32x4v cur[2][KC], nxt[2][KC];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) cur[i][ks] = nxt[i][ks] = (u32x4v){0u, 0u, 0u, 0u};
    // the first item is requested before the weights are staged: its latency runs under the LDS fill
    if (pair < p.npairs) load_item(pair, 0, cur);

    // ---- weights -> LDS in fragment order; bias in channel order ------------------------------------------------
    for (int f = wave; f < NCH * 4 * KS; f += 4) {
        const int T = f / KS, ks = f - T * KS;
        const int n = n0 + (T >> 2) * 64 + chunk_channel(T & 3, px);
        const uint4 v = *(const uint4*)(p.Wt + (long)n * p.ldw + ks * 32 + g * 8);
        *(uint4*)(wsm + f * 1024 + lane * 16) = v;
    }
    if (tid < NCH * 64) {
        const int n = n0 + tid;
        bias_s[tid] = p.bias ? p.bias[EPI == SE_CONVT ? n % p.Co : n] : 0.f;
    }
    __syncthreads();
    // The first item's registers are made "used" here, which puts the wait for its loads in front of the loop.  Without
    // it the loop header inherits "loads pending on cur" from this path, and the compiler's counter bookkeeping then
    // waits, in every iteration, for the loads it has just issued for the NEXT item: no prefetch at all.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) asm volatile("" : "+v"(cur[i][ks]));

    // Two register sets with a copy per item, and the prefetch skipped (not clamped) past the last item: a copy-free
    // ping-pong between the sets with an unconditional prefetch was measured 3-5 us SLOWER per launch -- the compiler
    // then protects the address/data registers of stores still in flight with s_waitcnt vmcnt(2) at the top of each
    // half.
    f32x4 acc[4][2];
    for (; pair < p.npairs; pair += nw) {
        // the weight fragments are loop-invariant LDS reads: left alone the compiler keeps ALL of them in registers (up
        // to 128 VGPRs) and the kernel drops to one wave per SIMD.  An offset it cannot see through keeps them in the loop.
        int woff = lane * 16;
        asm volatile("" : "+v"(woff));
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            if (kc + 1 < NKC) load_item(pair, kc + 1, nxt);
            else if (pair + nw < p.npairs) load_item(pair + nw, 0, nxt);
#pragma unroll 1
            for (int ch = 0; ch < NCH; ++ch) {
                if (kc == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 b4 = *(const float4*)(bias_s + ch * 64 + (j >> 1) * 32 + g * 8 + (j & 1) * 4);
                        acc[j][0] = acc[j][1] = (f32x4){b4.x, b4.y, b4.z, b4.w};
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ks = 0; ks < KC; ++ks) {
                        const bf16x8 wf = *(const bf16x8*)(wsm + ((ch * 4 + j) * KS + kc * KC + ks) * 1024 + woff);
                        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, cur[0][ks]), acc[j][0], 0, 0, 0);
                        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, cur[1][ks]), acc[j][1], 0, 0, 0);
                    }
                if (kc == NKC - 1) {
                    // lane: pixel px of fragment i, channels chunk*64 + {g*8 .. g*8+7} (tiles 0, 1) and {32 + g*8 ..} (tiles 2, 3)
                    const int nb = n0 + ch * 64;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int m = pair * 32 + i * 16 + px;
                        long orow;
                        if (EPI == SE_CONVT) {
                            const int w = m & (p.W - 1), h = (m >> p.wshift) & (p.H - 1), img = m >> (p.wshift + p.hshift);
                            const int tap = nb / p.Co, co0 = nb - tap * p.Co;
                            const long pix = ((long)img * 2 * p.H + 2 * h + (tap >> 1)) * (2 * p.W) + 2 * w + (tap & 1);
                            orow = pix * p.Co + co0;
                        } else {
                            orow = (long)m * p.ldo + nb;
                        }
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const f32x4 a = acc[hf * 2][i], b = acc[hf * 2 + 1][i];
                            if (EPI == SE_BNACT) {
                                // what the separate BatchNorm kernel computes from the stored conv output: the value
                                // rounded to bf16 first, then mul, add (two roundings), ReLU
                                float o8[8];
                                unsigned bits = 0u;
#pragma unroll
                                for (int r = 0; r < 8; ++r) {
                                    const float y = bf2f(f2bf(r < 4 ? a[r] : b[r - 4]));
                                    const float t = y * bsc[hf * 8 + r] + bsh[hf * 8 + r];
                                    const bool pos = t > 0.f;
                                    o8[r] = pos ? t : 0.f;
                                    bits |= pos ? (1u << r) : 0u;
                                }
                                *(uint4*)(p.out + orow + hf * 32 + g * 8) =
                                    make_uint4(pack2bf(o8[0], o8[1]), pack2bf(o8[2], o8[3]), pack2bf(o8[4], o8[5]), pack2bf(o8[6], o8[7]));
                                if (p.relu_bits) p.relu_bits[(long)m * (p.N >> 3) + ((nb + hf * 32) >> 3) + g] = (unsigned char)bits;
                                continue;
                            }
                            if (want_stats) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    ssum[hf * 8 + r] += a[r];       ssq[hf * 8 + r] += a[r] * a[r];
                                    ssum[hf * 8 + 4 + r] += b[r];   ssq[hf * 8 + 4 + r] += b[r] * b[r];
                                }
                            }
                            if (EPI == SE_STATS) continue;
                            const uint4 o = make_uint4(pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3]));
                            // plain stores: the consumer (BatchNorm apply) reads this tensor next, and at these sizes much
                            // of it is still in the memory-side cache (non-temporal stores measured 5-40 % slower here)
                            *(uint4*)(p.out + orow + hf * 32 + g * 8) = o;
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int ks = 0; ks < KC; ++ks) cur[i][ks] = nxt[i][ks];
        }
    }

    if (want_stats) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { ssum[e] = row16_sum_s(ssum[e]); ssq[e] = row16_sum_s(ssq[e]); }
        if (px == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = (e >> 3) * 32 + g * 8 + (e & 7);
                red[wave][0][c] = ssum[e];
                red[wave][1][c] = ssq[e];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            const float s = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
            // the NCH chunks of this workgroup are folded into one: a row has N / NCH columns, 64 per workgroup along y
            float* dst = which ? p.stat_sq : p.stat_sum;
            dst[(long)blockIdx.x * (p.N / NCH) + blockIdx.y * 64 + c] = s;
        }
    }
}

static int ilog2s(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// workgroups along x: enough resident waves to cover the memory latency, at most one pair of pixel fragments short
static int stream_grid_x(long M, int lds_bytes) {
    const long npairs = M / 32;
    int per_cu = (160 * 1024) / (lds_bytes + 3072);
    if (per_cu > 3) per_cu = 3;
    if (per_cu < 1) per_cu = 1;
    long gx = 256L * per_cu;
    const long need = (npairs + 3) / 4;
    if (gx > need) gx = need;
    return (int)gx;
}

template <int AMODE, int EPI, int KS, int NCH>
static int stream_launch(StreamGemm p, const char* name, double flops, double bytes, hipStream_t s) {
    p.wshift = ilog2s(p.W);
    p.hshift = ilog2s(p.H);
    p.npairs = p.M / 32;
    dim3 grid(stream_grid_x(p.M, NCH * 4 * KS * 1024), p.N / (64 * NCH));
    ProfScope prof(name, flops, bytes, s);
    hipLaunchKernelGGL((stream_gemm_kernel<AMODE, EPI, KS, NCH>), grid, dim3(256), 0, s, p);
    VAE_LAUNCH_CHECK(name);
    return 0;
}

// ---- first conv ---------------------------------------------------------------------------------------------------
bool stream_c8_ok(int n_img, int H, int W, int Cout) {
    return g_stream_gemm_mode != 1 && ilog2s(H) >= 0 && ilog2s(W) >= 0 && Cout % 64 == 0 && ((long)n_img * H * W) % 32 == 0 &&
           (long)n_img * H * W * Cout * 2 < (1L << 32);
}
int stream_c8_rows(int n_img, int H, int W) { return stream_grid_x((long)n_img * H * W, 4 * 3 * 1024); }

// y == null: statistics only.  scale / shift set: the activation relu(bf16(conv) * scale + shift) is stored in y's place
// (and relu_bits, if given); no statistics then.
int stream_c8_launch(const void* x8, const void* w_packed, const float* bias, void* y, float* stat_sum, float* stat_sq,
                     const float* scale, const float* shift, void* relu_bits, int n_img, int H, int W, int Cout, double flops,
                     double bytes, hipStream_t s) {
    StreamGemm p = {};
    p.A = (const bf16_t*)x8; p.Wt = (const bf16_t*)w_packed; p.bias = bias; p.out = (bf16_t*)y;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.scale = scale; p.shift = shift; p.relu_bits = (unsigned char*)relu_bits;
    p.M = n_img * H * W; p.N = Cout; p.ldw = 128; p.ldo = Cout; p.H = H; p.W = W;
    p.a_bytes = (unsigned)((long)p.M * 16);
    if (scale) return stream_launch<SA_C8, SE_BNACT, 3, 1>(p, "stream_gemm_kernel<c8 + bn + relu>", flops, bytes, s);
    if (!y) return stream_launch<SA_C8, SE_STATS, 3, 1>(p, "stream_gemm_kernel<c8 statistics>", flops, bytes, s);
    return stream_launch<SA_C8, SE_BF16, 3, 1>(p, "stream_gemm_kernel<c8>", flops, bytes, s);
}

// ---- ConvTranspose2d k2 s2 forward --------------------------------------------------------------------------------
// all four taps in one workgroup (the input is read once) when K fits the registers and a tap is one 64-channel chunk
// taps (= 64-channel chunks when Cout is 64) worked per workgroup on the same input registers: all four when the
// weights fit (static LDS: 64 KB), two for Cin = 128
static int convt_fwd_nch(int Cin, int Cout) { return Cout != 64 ? 1 : Cin == 64 ? 4 : Cin == 128 ? 2 : 1; }

// Measured against the tiled GEMM at batch 512: 64 -> 64 at 16x16 (67 MB out) 19 vs 27 us; 128 -> 64 at 8x8 (34 MB) a tie;
// 256 -> 128 at 4x4 (17 MB) 10.5 vs 9 us; the dgrad (reads the large tensor, writes the small one) never faster.  So:
// from 48 MB of output up, with one tap per 64-channel chunk (mode 2 lifts the size limit, for tests).
bool stream_convt_fwd_ok(int n_img, int H, int W, int Cin, int Cout) {
    const long out_bytes = (long)n_img * H * W * 4 * Cout * 2;
    return g_stream_gemm_mode != 1 && ilog2s(H) >= 0 && ilog2s(W) >= 0 && (Cin == 64 || Cin == 128 || Cin == 256) &&
           Cout % 64 == 0 && ((long)n_img * H * W) % 32 == 0 && out_bytes < (1L << 32) &&
           (g_stream_gemm_mode == 2 || (out_bytes >= (48L << 20) && Cout == 64 && Cin <= 128));
}
// column groups of Cout in a statistics row (the taps a workgroup does not fold itself)
int stream_convt_fwd_fold(int Cin, int Cout) { return 4 / convt_fwd_nch(Cin, Cout); }
int stream_convt_fwd_rows(int n_img, int H, int W, int Cin, int Cout) {
    const int nch = convt_fwd_nch(Cin, Cout);
    return stream_grid_x((long)n_img * H * W, nch * 4 * (Cin / 32) * 1024);
}

int stream_convt_fwd_launch(const void* x, const void* w_packed, const float* bias, void* y, float* stat_sum, float* stat_sq,
                            int n_img, int H, int W, int Cin, int Cout, double flops, double bytes, hipStream_t s) {
    StreamGemm p = {};
    p.A = (const bf16_t*)x; p.Wt = (const bf16_t*)w_packed; p.bias = bias; p.out = (bf16_t*)y;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.M = n_img * H * W; p.N = 4 * Cout; p.ldw = Cin; p.lda = Cin; p.H = H; p.W = W; p.Co = Cout;
    p.a_bytes = (unsigned)((long)p.M * Cin * 2);
    const char* name = "stream_gemm_kernel<convt_fwd>";
    const int nch = convt_fwd_nch(Cin, Cout);
    if (nch == 4) return stream_launch<SA_PLAIN, SE_CONVT, 2, 4>(p, name, flops, bytes, s);
    if (nch == 2) return stream_launch<SA_PLAIN, SE_CONVT, 4, 2>(p, name, flops, bytes, s);
    if (Cin == 64) return stream_launch<SA_PLAIN, SE_CONVT, 2, 1>(p, name, flops, bytes, s);
    if (Cin == 128) return stream_launch<SA_PLAIN, SE_CONVT, 4, 1>(p, name, flops, bytes, s);
    return stream_launch<SA_PLAIN, SE_CONVT, 8, 1>(p, name, flops, bytes, s);
}

}  // namespace vae
