// fp32 execution mode: see f32_path.h.  Every layer of the model that is a sum of products runs through ONE tiled
// kernel (64 x 64 outputs per workgroup, 16-deep K steps through LDS, one 32 x 32 block per wave on the fp32 matrix
// instruction) whose operands are fetched by per-mode index maps straight from the NCHW tensors and the parameters'
// own layouts; products are exact; the forward modes sum them in fp64 (the fp64 matrix instruction), the gradient
// modes in fp32 over blocks of 2048 values of k and in fp64 across the blocks.
#include "f32_path.h"
#include "prof.h"

namespace vae {
namespace f32 {

struct G32 {
    const float* a;
    const float* b;
    const float* bias;
    float* out;
    int M, N, K, kchunk;
    int n_img, H, W, Ci, Co, ks, pad;
    int epi, accumulate;
    double* slab;      // split-K partial sums (null: one split, results go straight to `out`)
};

// which index of an operand tile runs fastest in memory: the loads of a tile are spread over the threads accordingly
template <int MODE> struct Traits;
template <> struct Traits<CONV_FWD> { static constexpr bool a_kfast = false, b_kfast = true; };
template <> struct Traits<CONV_DGRAD> { static constexpr bool a_kfast = false, b_kfast = true; };
template <> struct Traits<CONV_WGRAD> { static constexpr bool a_kfast = true, b_kfast = true; };
template <> struct Traits<CONVT_FWD> { static constexpr bool a_kfast = false, b_kfast = false; };
template <> struct Traits<CONVT_DGRAD> { static constexpr bool a_kfast = false, b_kfast = true; };
template <> struct Traits<CONVT_WGRAD> { static constexpr bool a_kfast = true, b_kfast = true; };
template <> struct Traits<LIN_FWD> { static constexpr bool a_kfast = true, b_kfast = true; };
template <> struct Traits<LIN_DGRAD> { static constexpr bool a_kfast = true, b_kfast = false; };
template <> struct Traits<LIN_WGRAD> { static constexpr bool a_kfast = false, b_kfast = false; };

struct Pix { int n, h, w; };
__device__ __forceinline__ Pix pix_of(const G32& g, int p) {
    const int hw = g.H * g.W;
    Pix q;
    q.n = p / hw;
    const int r = p - q.n * hw;
    q.h = r / g.W;
    q.w = r - q.h * g.W;
    return q;
}
__device__ __forceinline__ float at4(const float* t, int n, int C, int c, int H, int W, int h, int w) {
    if ((unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) return 0.f;
    return t[((n * C + c) * H + h) * W + w];          // 32-bit index: the launcher checks that every tensor has < 2^31 elements
}
// k -> (channel, kernel row, kernel column) of a KS x KS kernel; KS is a compile-time constant, so the divisions
// are multiplications
template <int KS>
__device__ __forceinline__ void tap_of(int k, int& c, int& kh, int& kw) {
    constexpr int KS2 = KS * KS;
    c = k / KS2;
    const int t = k - c * KS2;
    kh = t / KS;
    kw = t - kh * KS;
}

// An operand tile is fetched element by element through these maps.  The index that is a PIXEL (m of the forward /
// input-gradient modes, k of the weight-gradient modes) is decomposed into (image, row, column) by the caller, once per
// thread and tile row where it is the tile's row index: `pm` is that decomposition when the mode's m is a pixel.
template <int MODE, int KS>
__device__ __forceinline__ float load_a(const G32& g, int m, int k, const Pix& pm) {
    if constexpr (MODE == CONV_FWD) {
        int ci, kh, kw;
        tap_of<KS>(k, ci, kh, kw);
        return at4(g.a, pm.n, g.Ci, ci, g.H, g.W, pm.h + kh - g.pad, pm.w + kw - g.pad);
    } else if constexpr (MODE == CONV_DGRAD) {
        int co, kh, kw;
        tap_of<KS>(k, co, kh, kw);
        return at4(g.a, pm.n, g.Co, co, g.H, g.W, pm.h + g.pad - kh, pm.w + g.pad - kw);
    } else if constexpr (MODE == CONV_WGRAD) {
        const Pix p = pix_of(g, k);
        return g.a[((p.n * g.Co + m) * g.H + p.h) * g.W + p.w];
    } else if constexpr (MODE == CONVT_FWD) {
        return g.a[((pm.n * g.Ci + k) * g.H + pm.h) * g.W + pm.w];
    } else if constexpr (MODE == CONVT_DGRAD) {
        const int co = k >> 2, kh = (k >> 1) & 1, kw = k & 1;
        return g.a[((pm.n * g.Co + co) * (2 * g.H) + 2 * pm.h + kh) * (2 * g.W) + 2 * pm.w + kw];
    } else if constexpr (MODE == CONVT_WGRAD) {
        const Pix p = pix_of(g, k);
        return g.a[((p.n * g.Ci + m) * g.H + p.h) * g.W + p.w];
    } else if constexpr (MODE == LIN_FWD || MODE == LIN_DGRAD) {
        return g.a[m * g.K + k];
    } else {   // LIN_WGRAD: dy [K rows][M]
        return g.a[k * g.M + m];
    }
}

template <int MODE, int KS>
__device__ __forceinline__ float load_b(const G32& g, int k, int n) {
    if constexpr (MODE == CONV_FWD) {
        return g.b[n * g.K + k];
    } else if constexpr (MODE == CONV_DGRAD) {
        constexpr int KS2 = KS * KS;
        const int co = k / KS2, t = k - co * KS2;
        return g.b[(co * g.Ci + n) * KS2 + t];
    } else if constexpr (MODE == CONV_WGRAD) {
        const Pix p = pix_of(g, k);
        int ci, kh, kw;
        tap_of<KS>(n, ci, kh, kw);
        return at4(g.b, p.n, g.Ci, ci, g.H, g.W, p.h + kh - g.pad, p.w + kw - g.pad);
    } else if constexpr (MODE == CONVT_FWD) {
        return g.b[k * g.N + n];
    } else if constexpr (MODE == CONVT_DGRAD) {
        return g.b[n * g.K + k];
    } else if constexpr (MODE == CONVT_WGRAD) {
        const Pix p = pix_of(g, k);
        const int co = n >> 2, kh = (n >> 1) & 1, kw = n & 1;
        return g.b[((p.n * g.Co + co) * (2 * g.H) + 2 * p.h + kh) * (2 * g.W) + 2 * p.w + kw];
    } else if constexpr (MODE == LIN_FWD) {
        return g.b[n * g.K + k];
    } else {   // LIN_DGRAD: weight [K][N]; LIN_WGRAD: x [K rows][N]
        return g.b[k * g.N + n];
    }
}

__device__ __forceinline__ float epilogue(float v, int epi) {
    if (epi == EPI_TANH) return tanhf(v);
    if (epi == EPI_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

template <int MODE>
__device__ __forceinline__ void store_c(const G32& g, int m, int n, double acc, int z) {
    if constexpr (MODE == CONV_FWD) {
        const Pix p = pix_of(g, m);
        const float v = (float)(acc + (g.bias ? (double)g.bias[n] : 0.0));
        g.out[(((long)p.n * g.Co + n) * g.H + p.h) * g.W + p.w] = epilogue(v, g.epi);
    } else if constexpr (MODE == CONV_DGRAD || MODE == CONVT_DGRAD) {
        const Pix p = pix_of(g, m);
        float* o = g.out + (((long)p.n * g.Ci + n) * g.H + p.h) * g.W + p.w;
        *o = g.accumulate ? (float)((double)*o + acc) : (float)acc;
    } else if constexpr (MODE == CONVT_FWD) {
        const Pix p = pix_of(g, m);
        const int co = n >> 2, kh = (n >> 1) & 1, kw = n & 1;
        const float v = (float)(acc + (g.bias ? (double)g.bias[co] : 0.0));
        g.out[(((long)p.n * g.Co + co) * (2 * g.H) + 2 * p.h + kh) * (2 * g.W) + 2 * p.w + kw] = epilogue(v, g.epi);
    } else if constexpr (MODE == LIN_FWD) {
        const float v = (float)(acc + (g.bias ? (double)g.bias[n] : 0.0));
        g.out[(long)m * g.N + n] = epilogue(v, g.epi);
    } else if constexpr (MODE == LIN_DGRAD) {
        float* o = g.out + (long)m * g.N + n;
        *o = g.accumulate ? (float)((double)*o + acc) : (float)acc;
    } else {   // weight gradients: [M][N] is the parameter's own layout
        g.out[(long)m * g.N + n] = (float)acc;
    }
}

constexpr int TM = 64, TN = 64, TK = 16, LDT = 68;     // tile; LDS row stride (floats)
constexpr int EPT = TM * TK / 256;                     // operand elements a thread fetches per tile and step

template <int MODE> struct MIsPixel { static constexpr bool value = MODE == CONV_FWD || MODE == CONV_DGRAD || MODE == CONVT_FWD || MODE == CONVT_DGRAD; };

// Lanes of the matrix instruction's result run along the index that is contiguous in the output tensor: the pixel (m)
// for the NCHW activations of the forward / input-gradient modes, the column (n) for row-major results.
template <int MODE> struct LanesAlongM { static constexpr bool value = MIsPixel<MODE>::value; };

// Which matrix instruction sums the products.
//   Forward modes: v_mfma_f64_16x16x4_f64 on operands widened to fp64 -- products of fp32 values are exact in fp64 and
//   the sum carries ~1e-16 per step, so a result is the correctly rounded fp32 value of the exact sum (but for ties
//   in the 29 bits below).  The forward pass needs that: its values decide ReLU, max-pool and sign(recon - x)
//   routes, and an activation that sits within fp32 summation noise of such a decision flips gradients in front of it
//   by 1e-3.  The exact sum is what every order of fp32 summation scatters around, so it takes the decision most
//   fp32 runs of the reference take.
//   Backward modes (input and weight gradients): v_mfma_f32_32x32x2_f32 -- exact fp32 products accumulated in fp32
//   (bitwise an fmaf chain) at twice the fp64 instruction's rate.  Every KBLOCK values of k the fp32 accumulators are
//   added into fp64 ones and cleared, so the rounding error does not grow with K (5e5 in a weight gradient): a
//   block's fp32 sum is off by ~1e-7 of sum |a b|, and the blocks combine in fp64 in a fixed order.  Nothing decides
//   anything on these values: the error stays ~1e-6 relative, inside the 2e-5 figure on gradients.
template <int MODE> struct ExactSum { static constexpr bool value = MODE == CONV_FWD || MODE == CONVT_FWD || MODE == LIN_FWD; };
constexpr int KBLOCK = 2048;

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) double f64x4;

template <int MODE, int KS>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const G32 g) {
    __shared__ __attribute__((aligned(16))) float As[TK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TK][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                     // wave tile: rows [32 wm, +32) x columns [32 wn, +32)
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN, z = blockIdx.z;
    const int kbeg = z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    // the tile rows / columns this thread fetches are the same in every K step: (mm[i], kk[i]) of A, (nn[i], kb[i]) of B
    int mm[EPT], ka[EPT], nn[EPT], kb[EPT];
    Pix pm[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        if (Traits<MODE>::a_kfast) { ka[i] = tid % TK; mm[i] = tid / TK + (256 / TK) * i; }
        else { mm[i] = tid & 63; ka[i] = (tid >> 6) + 4 * i; }
        if (Traits<MODE>::b_kfast) { kb[i] = tid % TK; nn[i] = tid / TK + (256 / TK) * i; }
        else { nn[i] = tid & 63; kb[i] = (tid >> 6) + 4 * i; }
        pm[i] = Pix{0, 0, 0};
        if (MIsPixel<MODE>::value && m0 + mm[i] < g.M) pm[i] = pix_of(g, m0 + mm[i]);
    }
    constexpr bool LM = LanesAlongM<MODE>::value, EXACT = ExactSum<MODE>::value;
    // fp32 instruction: one 32 x 32 block per wave, lane -> (row / column lane & 31, k = lane >> 5)
    f32x16 acc;
    double accd[16];
    // fp64 instruction: 2 x 2 blocks of 16 x 16 per wave, lane -> (row / column lane & 15, k = lane >> 4)
    f64x4 dacc[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accd[r] = 0.0; }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dacc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const float* ap = EXACT ? &As[lane >> 4][wm * 32 + (lane & 15)] : &As[lane >> 5][wm * 32 + (lane & 31)];
    const float* bp = EXACT ? &Bs[lane >> 4][wn * 32 + (lane & 15)] : &Bs[lane >> 5][wn * 32 + (lane & 31)];
    int since = 0;
    float av[EPT], bv[EPT];
    auto fetch = [&](int k0) {              // all the loads of a step are requested together
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            av[i] = (m0 + mm[i] < g.M && k0 + ka[i] < kend) ? load_a<MODE, KS>(g, m0 + mm[i], k0 + ka[i], pm[i]) : 0.f;
            bv[i] = (n0 + nn[i] < g.N && k0 + kb[i] < kend) ? load_b<MODE, KS>(g, k0 + kb[i], n0 + nn[i]) : 0.f;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
#pragma unroll
        for (int i = 0; i < EPT; ++i) {
            As[ka[i]][mm[i]] = av[i];
            Bs[kb[i]][nn[i]] = bv[i];
        }
        __syncthreads();
        if (k0 + TK < kend) fetch(k0 + TK);     // the next step's operands travel while this step is multiplied
        if constexpr (EXACT) {
#pragma unroll
            for (int kk = 0; kk < TK / 4; ++kk) {
                const double a[2] = {(double)ap[4 * kk * LDT], (double)ap[4 * kk * LDT + 16]};
                const double b[2] = {(double)bp[4 * kk * LDT], (double)bp[4 * kk * LDT + 16]};
                // result[i][j] = sum_k first[i][k] * second[k][j]; lanes run along j
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        dacc[i][j] = LM ? __builtin_amdgcn_mfma_f64_16x16x4f64(b[i], a[j], dacc[i][j], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], dacc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < TK / 2; ++kk) {
                const float a = ap[2 * kk * LDT], b = bp[2 * kk * LDT];
                acc = LM ? __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            since += TK;
            if (since >= KBLOCK) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { accd[r] += (double)acc[r]; acc[r] = 0.f; }
                since = 0;
            }
        }
        __syncthreads();
    }
    // one split: the result itself; several: this split's partial sums (fp64, the index the lanes run along fastest)
    auto put = [&](int m, int n, double v) {
        if (m >= g.M || n >= g.N) return;
        if (g.slab) g.slab[(long)z * g.M * g.N + (LM ? (long)n * g.M + m : (long)m * g.N + n)] = v;
        else store_c<MODE>(g, m, n, v, 0);
    };
    if constexpr (EXACT) {
        // register r of block (bi, bj): row 16 bi + (lane >> 4) + 4 r, column 16 bj + (lane & 15)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int bj = 0; bj < 2; ++bj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * bi + (lane >> 4) + 4 * r, j = 16 * bj + (lane & 15);
                    put(m0 + wm * 32 + (LM ? j : i), n0 + wn * 32 + (LM ? i : j), dacc[bi][bj][r]);
                }
    } else {
        // register r of the 32 x 32 result: row i = 8 (r / 4) + 4 (lane / 32) + r % 4, column j = lane % 32
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3), j = lane & 31;
            put(m0 + wm * 32 + (LM ? j : i), n0 + wn * 32 + (LM ? i : j), accd[r] + (double)acc[r]);
        }
    }
}

// result element (m, n) = sum over the splits of their partial sums, in split order, in fp64, then the mode's own
// epilogue and store.  Threads run along the index that is fastest in the slabs (and in the output).
template <int MODE>
__global__ __launch_bounds__(256) void fold_store_kernel(const G32 g, int nsplit) {
    constexpr bool LM = LanesAlongM<MODE>::value;
    const long total = (long)g.M * g.N;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    // four independent chains keep four loads in flight (one chain is a latency ladder of `nsplit` round trips);
    // the grouping is fixed, so the sum is reproducible
    double s4[4] = {0.0, 0.0, 0.0, 0.0};
    int z = 0;
    for (; z + 3 < nsplit; z += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] += g.slab[(long)(z + u) * total + i];
    }
    for (; z < nsplit; ++z) s4[0] += g.slab[(long)z * total + i];
    const double s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    const int m = LM ? (int)(i % g.M) : (int)(i / g.N), n = LM ? (int)(i / g.M) : (int)(i % g.N);
    store_c<MODE>(g, m, n, s, 0);
}

static void mnk(int mode, int n_img, int H, int W, int Ci, int Co, int ks, long* M, long* N, long* K) {
    const long pix = (long)n_img * H * W;
    switch (mode) {
        case CONV_FWD: *M = pix; *N = Co; *K = (long)Ci * ks * ks; break;
        case CONV_DGRAD: *M = pix; *N = Ci; *K = (long)Co * ks * ks; break;
        case CONV_WGRAD: *M = Co; *N = (long)Ci * ks * ks; *K = pix; break;
        case CONVT_FWD: *M = pix; *N = 4L * Co; *K = Ci; break;
        case CONVT_DGRAD: *M = pix; *N = Ci; *K = 4L * Co; break;
        case CONVT_WGRAD: *M = Ci; *N = 4L * Co; *K = pix; break;
        case LIN_FWD: *M = n_img; *N = Co; *K = Ci; break;
        case LIN_DGRAD: *M = n_img; *N = Ci; *K = Co; break;
        default: *M = Co; *N = Ci; *K = n_img; break;     // LIN_WGRAD
    }
}

// Split the K range so that about 1024 workgroups run when the output has few tiles (weight gradients: K is the pixel
// count; at small batches also the 4x4 / 8x8 layers and the Linear layers, whose 2-64 output tiles would each walk
// thousands of k alone).  Chunks are multiples of the K step of at least 128.
static int k_splits(long M, long N, long K, int* kchunk) {
    const long tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    long want = tiles >= 512 ? 1 : (1024 + tiles - 1) / tiles;
    const long most = (K + 127) / 128;
    if (want > most) want = most;
    if (want < 1) want = 1;
    long chunk = ((K + want - 1) / want + TK - 1) / TK * TK;
    *kchunk = (int)chunk;
    return (int)((K + chunk - 1) / chunk);
}

// scratch for the fp64 partial sums of a split GEMM, in floats (two per partial sum)
long slab_need(int mode, int n_img, int H, int W, int Ci, int Co, int ks) {
    long M, N, K;
    mnk(mode, n_img, H, W, Ci, Co, ks, &M, &N, &K);
    int kchunk;
    const int ns = k_splits(M, N, K, &kchunk);
    return ns > 1 ? 2L * ns * M * N : 0;
}

int gemm(int mode, const Gemm& q, hipStream_t s) {
    VAE_CHECK(mode >= CONV_FWD && mode <= LIN_WGRAD, "f32 gemm: unknown mode %d", mode);
    VAE_CHECK(q.a && q.b && q.out, "f32 gemm: null pointer (mode %d)", mode);
    VAE_CHECK(q.n_img > 0 && q.H > 0 && q.W > 0 && q.Ci > 0 && q.Co > 0, "f32 gemm: bad geometry n=%d H=%d W=%d Ci=%d Co=%d",
              q.n_img, q.H, q.W, q.Ci, q.Co);
    const bool conv = mode <= CONV_WGRAD;
    VAE_CHECK(!conv || q.ks == 1 || q.ks == 3, "f32 gemm: kernel size %d (1 or 3)", q.ks);
    long M, N, K;
    mnk(mode, q.n_img, q.H, q.W, q.Ci, q.Co, q.ks, &M, &N, &K);
    VAE_CHECK(M < (1L << 31) && N < (1L << 31) && K < (1L << 31), "f32 gemm: a dimension exceeds 2^31");
    // the operand index maps run in 32-bit arithmetic: no operand tensor may reach 2^31 elements (the largest
    // activation is [n][max(Ci, Co)][2H][2W], a ConvTranspose's output; the largest weight Co x Ci x 9)
    VAE_CHECK((long)q.n_img * q.H * q.W * 4 * (q.Ci > q.Co ? q.Ci : q.Co) < (1L << 31) && (long)q.Ci * q.Co * 9 < (1L << 31),
              "f32 gemm: an operand has 2^31 elements or more (n=%d H=%d W=%d Ci=%d Co=%d)", q.n_img, q.H, q.W, q.Ci, q.Co);
    G32 g = {};
    g.a = q.a; g.b = q.b; g.bias = q.bias; g.out = q.out;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.n_img = q.n_img; g.H = q.H; g.W = q.W; g.Ci = q.Ci; g.Co = q.Co; g.ks = conv ? q.ks : 1; g.pad = g.ks / 2;
    g.epi = q.epi; g.accumulate = q.accumulate;
    const int splits = k_splits(M, N, K, &g.kchunk);
    g.slab = nullptr;
    if (splits > 1) {
        VAE_CHECK(q.slab, "f32 gemm: this geometry splits K %d ways and needs %ld floats of scratch", splits, 2L * splits * M * N);
        VAE_CHECK(((uintptr_t)q.slab & 7) == 0, "f32 gemm: the split-K scratch must be 8-byte aligned");
        g.slab = (double*)q.slab;
    }
    const dim3 grid((unsigned)((M + TM - 1) / TM), (unsigned)((N + TN - 1) / TN), (unsigned)splits);
    VAE_CHECK(grid.y < 65536 && grid.z < 65536, "f32 gemm: grid too large");
    const dim3 fgrid((unsigned)((M * N + 255) / 256));
    {
        static const char* const names[9] = {"gemm_f32_kernel<conv fwd>", "gemm_f32_kernel<conv dgrad>", "gemm_f32_kernel<conv wgrad>",
                                             "gemm_f32_kernel<convT fwd>", "gemm_f32_kernel<convT dgrad>", "gemm_f32_kernel<convT wgrad>",
                                             "gemm_f32_kernel<linear fwd>", "gemm_f32_kernel<linear dgrad>", "gemm_f32_kernel<linear wgrad>"};
        ProfScope prof(names[mode], 2.0 * (double)M * (double)N * (double)K, 0.0, s);
#define GO(MD, KS_)                                                                         \
        do {                                                                                \
            hipLaunchKernelGGL((gemm_f32_kernel<MD, KS_>), grid, dim3(256), 0, s, g);       \
            if (splits > 1) hipLaunchKernelGGL((fold_store_kernel<MD>), fgrid, dim3(256), 0, s, g, splits); \
        } while (0)
        switch (mode) {
            case CONV_FWD: if (g.ks == 3) GO(CONV_FWD, 3); else GO(CONV_FWD, 1); break;
            case CONV_DGRAD: if (g.ks == 3) GO(CONV_DGRAD, 3); else GO(CONV_DGRAD, 1); break;
            case CONV_WGRAD: if (g.ks == 3) GO(CONV_WGRAD, 3); else GO(CONV_WGRAD, 1); break;
            case CONVT_FWD: GO(CONVT_FWD, 1); break;
            case CONVT_DGRAD: GO(CONVT_DGRAD, 1); break;
            case CONVT_WGRAD: GO(CONVT_WGRAD, 1); break;
            case LIN_FWD: GO(LIN_FWD, 1); break;
            case LIN_DGRAD: GO(LIN_DGRAD, 1); break;
            default: GO(LIN_WGRAD, 1); break;
        }
#undef GO
        VAE_LAUNCH_CHECK("gemm_f32_kernel");
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- BatchNorm2d
__device__ __forceinline__ double block_sum_d(double v, double* red) {      // every thread gets the sum
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// Per-channel reductions over (image, position) run in two stages so that they fill the chip whatever the channel
// count: grid (C, S) -- workgroup (c, s) sums its contiguous share of the images in fp64 and leaves one partial per
// quantity -- then one thread per channel adds the S partials in split order.  (One workgroup per channel left 64-256
// workgroups walking the whole batch: a quarter of the fp32 step.)
constexpr int MAX_CHAN_SPLIT = 64;
long chan_part_doubles(int C) { return 2L * (C > 1024 ? C : 1024) + 2L * 256; }
static int chan_splits(int n_img, int C) {
    int s = 1024 / C;
    if (s < 1) s = 1;
    if (s > n_img) s = n_img;
    return s > MAX_CHAN_SPLIT ? MAX_CHAN_SPLIT : s;
}
__device__ __forceinline__ void split_range(int n_img, int& n0, int& n1) {
    n0 = (int)((long)blockIdx.y * n_img / gridDim.y);
    n1 = (int)((long)(blockIdx.y + 1) * n_img / gridDim.y);
}

__global__ __launch_bounds__(256) void bn_stats_part_kernel(const float* __restrict__ y, int n_img, int C, int HW,
                                                            double* __restrict__ part) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    int n0, n1;
    split_range(n_img, n0, n1);
    double s = 0.0, q = 0.0;
    for (int n = n0; n < n1; ++n) {
        const float* p = y + ((long)n * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += 256) {
            const double v = (double)p[i];
            s += v;
            q = fma(v, v, q);
        }
    }
    s = block_sum_d(s, red);
    q = block_sum_d(q, red);
    if (threadIdx.x == 0) {
        part[((long)c * gridDim.y + blockIdx.y) * 2] = s;
        part[((long)c * gridDim.y + blockIdx.y) * 2 + 1] = q;
    }
}

__global__ __launch_bounds__(256) void bn_stats_fin_kernel(const double* __restrict__ part, int S, double cnt, int C,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* running_mean, float* running_var, int train, float momentum,
                                                           float eps, float* coef, int cs) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double mean, var;
    if (train) {
        double s = 0.0, q = 0.0;
        for (int i = 0; i < S; ++i) { s += part[((long)c * S + i) * 2]; q += part[((long)c * S + i) * 2 + 1]; }
        mean = s / cnt;
        var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const double unb = var * (cnt / (cnt > 1.0 ? cnt - 1.0 : 1.0));
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    } else {
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    }
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    coef[c] = sc;
    coef[cs + c] = beta[c] - (float)mean * sc;
    coef[2 * cs + c] = (float)mean;
    coef[3 * cs + c] = rstd;
}

int bn_stats(const float* y, int n_img, int C, int HW, const float* gamma, const float* beta, float* running_mean,
             float* running_var, int train, float momentum, float eps, float* coef, int coef_stride, double* part,
             hipStream_t s) {
    VAE_CHECK(y && gamma && beta && running_mean && running_var && coef && part, "f32 bn_stats: null pointer");
    VAE_CHECK(n_img > 0 && C > 0 && C <= coef_stride && HW > 0, "f32 bn_stats: bad geometry n=%d C=%d HW=%d", n_img, C, HW);
    const int S = chan_splits(n_img, C);
    if (train) {
        hipLaunchKernelGGL(bn_stats_part_kernel, dim3(C, S), dim3(256), 0, s, y, n_img, C, HW, part);
        VAE_LAUNCH_CHECK("bn_stats_part_kernel");
    }
    hipLaunchKernelGGL(bn_stats_fin_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, S, (double)n_img * HW, C, gamma, beta,
                       running_mean, running_var, train, momentum, eps, coef, coef_stride);
    VAE_LAUNCH_CHECK("bn_stats_fin_kernel");
    return 0;
}

// keep flag of element (n, c, h, w): the dropout streams are indexed by the NHWC element index (the layout of the
// bf16 path's tensors), so both precisions drop the same elements for the same seed
__device__ __forceinline__ bool keep_at(const DropCfg& d, long nhwc) {
    if (d.mask) return d.mask[nhwc] != 0;
    uint32_t w[4];
    philox_block(d.key, (uint32_t)(nhwc >> 3), w);
    const uint32_t draw = (w[(nhwc >> 1) & 3] >> (16 * (int)(nhwc & 1))) & 0xFFFFu;
    return draw >= d.thr;
}

// dropout(relu(y * scale + shift)) of one element
__device__ __forceinline__ float act_at(const float* y, float sc, float sh, const DropCfg& d, int n, int c, int h, int w, int C,
                                        int H, int W) {
    const float a = y[(((long)n * C + c) * H + h) * W + w] * sc + sh;
    float v = a > 0.f ? a : 0.f;
    if (d.enabled) v = keep_at(d, (((long)n * H + h) * W + w) * C + c) ? v * d.inv_keep : 0.f;
    return v;
}

__global__ __launch_bounds__(256) void bn_act_fwd_f32_kernel(const float* __restrict__ y, const float* __restrict__ coef, int cs,
                                                             float* __restrict__ out, unsigned char* __restrict__ argmax, int n_img,
                                                             int C, int H, int W, int pool, const DropCfg d) {
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    const long total = (long)n_img * C * Ho * Wo;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int wo = (int)(i % Wo), ho = (int)((i / Wo) % Ho), c = (int)((i / ((long)Wo * Ho)) % C), n = (int)(i / ((long)Wo * Ho * C));
    const float sc = coef[c], sh = coef[cs + c];
    if (!pool) {
        out[i] = act_at(y, sc, sh, d, n, c, ho, wo, C, H, W);
        return;
    }
    // nn.MaxPool2d(2): the first maximum in scan order wins
    float best = act_at(y, sc, sh, d, n, c, 2 * ho, 2 * wo, C, H, W);
    int arg = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const float v = act_at(y, sc, sh, d, n, c, 2 * ho + (k >> 1), 2 * wo + (k & 1), C, H, W);
        if (v > best) { best = v; arg = k; }
    }
    out[i] = best;
    if (argmax) argmax[i] = (unsigned char)arg;
}

int bn_act_fwd(const float* y, const float* coef, int coef_stride, float* out, unsigned char* argmax, int n_img, int C, int H,
               int W, int pool, const DropCfg& drop, hipStream_t s) {
    VAE_CHECK(y && coef && out, "f32 bn_act_fwd: null pointer");
    VAE_CHECK(n_img > 0 && C > 0 && H > 0 && W > 0, "f32 bn_act_fwd: bad geometry");
    VAE_CHECK(!pool || (H % 2 == 0 && W % 2 == 0), "f32 bn_act_fwd: pooling needs even H, W (got %dx%d)", H, W);
    const long total = (long)n_img * C * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    hipLaunchKernelGGL(bn_act_fwd_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, y, coef, coef_stride, out,
                       argmax, n_img, C, H, W, pool, drop);
    VAE_LAUNCH_CHECK("bn_act_fwd_f32_kernel");
    return 0;
}

// d loss / d (y * scale + shift) at one full-resolution element: the incoming gradient routed back through the pool
// (the window's winner takes it), the dropout scaling and the ReLU
__device__ __forceinline__ float dpre_at(const float* g, const float* y, float sc, float sh, const unsigned char* argmax,
                                         const DropCfg& d, int n, int c, int h, int w, int C, int H, int W, int pool) {
    const float yv = y[(((long)n * C + c) * H + h) * W + w];
    if (!(yv * sc + sh > 0.f)) return 0.f;
    float gv;
    if (pool) {
        const long gi = (((long)n * C + c) * (H / 2) + (h >> 1)) * (W / 2) + (w >> 1);
        if (argmax[gi] != (unsigned char)(((h & 1) << 1) | (w & 1))) return 0.f;
        gv = g[gi];
    } else {
        gv = g[(((long)n * C + c) * H + h) * W + w];
    }
    if (d.enabled) gv = keep_at(d, (((long)n * H + h) * W + w) * C + c) ? gv * d.inv_keep : 0.f;
    return gv;
}

__global__ __launch_bounds__(256) void bn_act_bwd_reduce_f32_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                    const float* __restrict__ coef, int cs,
                                                                    const unsigned char* __restrict__ argmax,
                                                                    double* __restrict__ part, int n_img, int C, int H, int W,
                                                                    int pool, const DropCfg d) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    int n0, n1;
    split_range(n_img, n0, n1);
    const float sc = coef[c], sh = coef[cs + c], mean = coef[2 * cs + c], rstd = coef[3 * cs + c];
    double s1 = 0.0, s2 = 0.0;
    for (int n = n0; n < n1; ++n)
        for (int i = threadIdx.x; i < H * W; i += 256) {
            const int h = i / W, w = i - h * W;
            const float dp = dpre_at(g, y, sc, sh, argmax, d, n, c, h, w, C, H, W, pool);
            if (dp != 0.f) {
                const float xh = (y[(((long)n * C + c) * H + h) * W + w] - mean) * rstd;
                s1 += (double)dp;
                s2 = fma((double)dp, (double)xh, s2);
            }
        }
    s1 = block_sum_d(s1, red);
    s2 = block_sum_d(s2, red);
    if (threadIdx.x == 0) {
        part[((long)c * gridDim.y + blockIdx.y) * 2] = s1;
        part[((long)c * gridDim.y + blockIdx.y) * 2 + 1] = s2;
    }
}

// out1[c] = sum of the first partials of channel c, out2[c] (optional) of the second; copies (optional) [C | C]
__global__ __launch_bounds__(256) void chan_fin_kernel(const double* __restrict__ part, int S, int C, int stride, float* out1,
                                                       float* out2, float* copies) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int i = 0; i < S; ++i) {
        a += part[((long)c * S + i) * stride];
        if (stride == 2) b += part[((long)c * S + i) * 2 + 1];
    }
    out1[c] = (float)a;
    if (out2) out2[c] = (float)b;
    if (copies) { copies[c] = (float)a; copies[C + c] = (float)b; }
}

__global__ __launch_bounds__(256) void bn_act_bwd_apply_f32_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                                   const float* __restrict__ coef, int cs,
                                                                   const float* __restrict__ gamma,
                                                                   const unsigned char* __restrict__ argmax,
                                                                   const float* __restrict__ sums, float* __restrict__ dy, int n_img,
                                                                   int C, int H, int W, int pool, const DropCfg d) {
    const long total = (long)n_img * C * H * W;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int w = (int)(i % W), h = (int)((i / W) % H), c = (int)((i / ((long)W * H)) % C), n = (int)(i / ((long)W * H * C));
    const float sc = coef[c], sh = coef[cs + c], mean = coef[2 * cs + c], rstd = coef[3 * cs + c];
    const float inv_cnt = 1.0f / ((float)n_img * (float)(H * W));
    const float dp = dpre_at(g, y, sc, sh, argmax, d, n, c, h, w, C, H, W, pool);
    const float xh = (y[i] - mean) * rstd;
    // (dy_hat - mean(dy_hat) - x_hat * mean(dy_hat * x_hat)) * gamma * rstd
    dy[i] = (dp - sums[c] * inv_cnt - xh * (sums[C + c] * inv_cnt)) * (gamma[c] * rstd);
}

int bn_act_bwd(const float* g, const float* y, const float* coef, int coef_stride, const float* gamma, const unsigned char* argmax,
               float* dy, float* dgamma, float* dbeta, float* sums, double* part, int n_img, int C, int H, int W, int pool,
               const DropCfg& drop, hipStream_t s) {
    VAE_CHECK(g && y && coef && gamma && dy && dgamma && dbeta && sums && part, "f32 bn_act_bwd: null pointer");
    VAE_CHECK(n_img > 0 && C > 0 && H > 0 && W > 0, "f32 bn_act_bwd: bad geometry");
    VAE_CHECK(!pool || argmax, "f32 bn_act_bwd: the pooled form needs the forward's argmax bytes");
    const int S = chan_splits(n_img, C);
    hipLaunchKernelGGL(bn_act_bwd_reduce_f32_kernel, dim3(C, S), dim3(256), 0, s, g, y, coef, coef_stride, argmax, part, n_img, C, H,
                       W, pool, drop);
    VAE_LAUNCH_CHECK("bn_act_bwd_reduce_f32_kernel");
    // d beta = sum(dy_hat), d gamma = sum(dy_hat * x_hat); the apply pass reads both from `sums`
    hipLaunchKernelGGL(chan_fin_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, S, C, 2, dbeta, dgamma, sums);
    VAE_LAUNCH_CHECK("chan_fin_kernel");
    const long total = (long)n_img * C * H * W;
    hipLaunchKernelGGL(bn_act_bwd_apply_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, y, coef, coef_stride,
                       gamma, argmax, sums, dy, n_img, C, H, W, pool, drop);
    VAE_LAUNCH_CHECK("bn_act_bwd_apply_f32_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------- small pieces
__global__ __launch_bounds__(256) void chan_sum_f32_kernel(const float* __restrict__ t, int n_img, int C, int HW,
                                                           double* __restrict__ part) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    int n0, n1;
    split_range(n_img, n0, n1);
    double s = 0.0;
    const long per = (long)(n1 - n0) * HW;
    for (long i = threadIdx.x; i < per; i += 256) {
        const long n = n0 + i / HW, r = i % HW;
        s += (double)t[(n * C + c) * HW + r];
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) part[(long)c * gridDim.y + blockIdx.y] = s;
}
int chan_sum(const float* t, int n_img, int C, int HW, float* out, double* part, hipStream_t s) {
    VAE_CHECK(t && out && part && n_img > 0 && C > 0 && HW > 0, "f32 chan_sum: bad arguments");
    const int S = chan_splits(n_img, C);
    hipLaunchKernelGGL(chan_sum_f32_kernel, dim3(C, S), dim3(256), 0, s, t, n_img, C, HW, part);
    VAE_LAUNCH_CHECK("chan_sum_f32_kernel");
    hipLaunchKernelGGL(chan_fin_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, S, C, 1, out, (float*)nullptr, (float*)nullptr);
    VAE_LAUNCH_CHECK("chan_fin_kernel");
    return 0;
}

__global__ __launch_bounds__(256) void act_bwd_f32_kernel(const float* __restrict__ dv, const float* __restrict__ v,
                                                          float* __restrict__ dpre, long n, int sigmoid) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float o = v[i];
    dpre[i] = sigmoid ? dv[i] * ((1.0f - o) * o) : dv[i] * (1.0f - o * o);
}
int tanh_bwd(const float* dh, const float* h, float* dpre, long n, hipStream_t s) {
    VAE_CHECK(dh && h && dpre && n > 0, "f32 tanh_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dh, h, dpre, n, 0);
    VAE_LAUNCH_CHECK("act_bwd_f32_kernel");
    return 0;
}
int sigmoid_bwd(const float* dr, const float* r, float* dpre, long n, hipStream_t s) {
    VAE_CHECK(dr && r && dpre && n > 0, "f32 sigmoid_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dr, r, dpre, n, 1);
    VAE_LAUNCH_CHECK("act_bwd_f32_kernel");
    return 0;
}

}  // namespace f32
}  // namespace vae
