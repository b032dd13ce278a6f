// fp32 execution mode: see f32_path.h.  Every layer of the model that is a sum of products runs through ONE tiled
// kernel (64 x 64 outputs per workgroup, 16-deep K steps through LDS, 4 x 4 outputs per thread) whose operands are
// fetched by per-mode index maps straight from the NCHW tensors and the parameters' own layouts; the products of
// fp32 operands are exact in fp64 and are summed there, so a result carries one fp32 rounding whatever K is.
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
    return t[(((long)n * C + c) * H + h) * W + w];
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
        return g.a[(((long)p.n * g.Co + m) * g.H + p.h) * g.W + p.w];
    } else if constexpr (MODE == CONVT_FWD) {
        return g.a[(((long)pm.n * g.Ci + k) * g.H + pm.h) * g.W + pm.w];
    } else if constexpr (MODE == CONVT_DGRAD) {
        const int co = k >> 2, kh = (k >> 1) & 1, kw = k & 1;
        return g.a[(((long)pm.n * g.Co + co) * (2 * g.H) + 2 * pm.h + kh) * (2 * g.W) + 2 * pm.w + kw];
    } else if constexpr (MODE == CONVT_WGRAD) {
        const Pix p = pix_of(g, k);
        return g.a[(((long)p.n * g.Ci + m) * g.H + p.h) * g.W + p.w];
    } else if constexpr (MODE == LIN_FWD || MODE == LIN_DGRAD) {
        return g.a[(long)m * g.K + k];
    } else {   // LIN_WGRAD: dy [K rows][M]
        return g.a[(long)k * g.M + m];
    }
}

template <int MODE, int KS>
__device__ __forceinline__ float load_b(const G32& g, int k, int n) {
    if constexpr (MODE == CONV_FWD) {
        return g.b[(long)n * g.K + k];
    } else if constexpr (MODE == CONV_DGRAD) {
        constexpr int KS2 = KS * KS;
        const int co = k / KS2, t = k - co * KS2;
        return g.b[((long)co * g.Ci + n) * KS2 + t];
    } else if constexpr (MODE == CONV_WGRAD) {
        const Pix p = pix_of(g, k);
        int ci, kh, kw;
        tap_of<KS>(n, ci, kh, kw);
        return at4(g.b, p.n, g.Ci, ci, g.H, g.W, p.h + kh - g.pad, p.w + kw - g.pad);
    } else if constexpr (MODE == CONVT_FWD) {
        return g.b[(long)k * g.N + n];
    } else if constexpr (MODE == CONVT_DGRAD) {
        return g.b[(long)n * g.K + k];
    } else if constexpr (MODE == CONVT_WGRAD) {
        const Pix p = pix_of(g, k);
        const int co = n >> 2, kh = (n >> 1) & 1, kw = n & 1;
        return g.b[(((long)p.n * g.Co + co) * (2 * g.H) + 2 * p.h + kh) * (2 * g.W) + 2 * p.w + kw];
    } else if constexpr (MODE == LIN_FWD) {
        return g.b[(long)n * g.K + k];
    } else {   // LIN_DGRAD: weight [K][N]; LIN_WGRAD: x [K rows][N]
        return g.b[(long)k * g.N + n];
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
    } else {   // weight gradients: split z of the K range -> its own [M][N] slab (or the gradient itself, one split)
        g.out[(long)z * g.M * g.N + (long)m * g.N + n] = (float)acc;
    }
}

constexpr int TM = 64, TN = 64, TK = 16, LDT = 68;     // tile; LDS row stride (floats)

template <int MODE> struct MIsPixel { static constexpr bool value = MODE == CONV_FWD || MODE == CONV_DGRAD || MODE == CONVT_FWD || MODE == CONVT_DGRAD; };

template <int MODE, int KS>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const G32 g) {
    // (tiles staged as fp64 -- one conversion per operand instead of one per product -- measured slower: 7.6 vs 10.3
    // TFLOP/s, the doubled LDS reads cost more than the conversions)
    __shared__ __attribute__((aligned(16))) float As[TK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[TK][LDT];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN, z = blockIdx.z;
    const int kbeg = z * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    // the tile rows / columns this thread fetches are the same in every K step: (mm[i], kk[i]) of A, (nn[i], kb[i]) of B
    int mm[4], ka[4], nn[4], kb[4];
    Pix pm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (Traits<MODE>::a_kfast) { ka[i] = tid & 15; mm[i] = (tid >> 4) + 16 * i; }
        else { mm[i] = tid & 63; ka[i] = (tid >> 6) + 4 * i; }
        if (Traits<MODE>::b_kfast) { kb[i] = tid & 15; nn[i] = (tid >> 4) + 16 * i; }
        else { nn[i] = tid & 63; kb[i] = (tid >> 6) + 4 * i; }
        pm[i] = Pix{0, 0, 0};
        if (MIsPixel<MODE>::value && m0 + mm[i] < g.M) pm[i] = pix_of(g, m0 + mm[i]);
    }
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = kbeg; k0 < kend; k0 += TK) {
        float av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // all eight loads of a step are requested before any is stored
            av[i] = (m0 + mm[i] < g.M && k0 + ka[i] < kend) ? load_a<MODE, KS>(g, m0 + mm[i], k0 + ka[i], pm[i]) : 0.f;
            bv[i] = (n0 + nn[i] < g.N && k0 + kb[i] < kend) ? load_b<MODE, KS>(g, k0 + kb[i], n0 + nn[i]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            As[ka[i]][mm[i]] = av[i];
            Bs[kb[i]][nn[i]] = bv[i];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            const float4 a4 = *(const float4*)&As[kk][ty * 4];
            const float4 b4 = *(const float4*)&Bs[kk][tx * 4];
            const double a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < g.M && n < g.N) store_c<MODE>(g, m, n, acc[i][j], z);
        }
}

// out[i] = sum over the splits of slab[z][i], in split order, in fp64
__global__ __launch_bounds__(256) void fold_f32_kernel(const float* __restrict__ slab, int nsplit, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int z = 0; z < nsplit; ++z) s += (double)slab[(long)z * n + i];
    out[i] = (float)s;
}

static bool is_wgrad(int mode) { return mode == CONV_WGRAD || mode == CONVT_WGRAD || mode == LIN_WGRAD; }

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

// split the K range of a weight gradient so that about 1024 workgroups run; chunks are multiples of 16 of at least 256
static int wgrad_splits(long M, long N, long K, int* kchunk) {
    const long tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    long want = (1024 + tiles - 1) / tiles;
    const long most = (K + 255) / 256;
    if (want > most) want = most;
    if (want < 1) want = 1;
    long chunk = ((K + want - 1) / want + 15) / 16 * 16;
    *kchunk = (int)chunk;
    return (int)((K + chunk - 1) / chunk);
}

long slab_need(int mode, int n_img, int H, int W, int Ci, int Co, int ks) {
    if (!is_wgrad(mode)) return 0;
    long M, N, K;
    mnk(mode, n_img, H, W, Ci, Co, ks, &M, &N, &K);
    int kchunk;
    const int ns = wgrad_splits(M, N, K, &kchunk);
    return ns > 1 ? (long)ns * M * N : 0;
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
    G32 g = {};
    g.a = q.a; g.b = q.b; g.bias = q.bias; g.out = q.out;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.n_img = q.n_img; g.H = q.H; g.W = q.W; g.Ci = q.Ci; g.Co = q.Co; g.ks = conv ? q.ks : 1; g.pad = g.ks / 2;
    g.epi = q.epi; g.accumulate = q.accumulate;
    int splits = 1;
    g.kchunk = (int)((K + 15) / 16 * 16);
    float* final_out = q.out;
    if (is_wgrad(mode)) {
        splits = wgrad_splits(M, N, K, &g.kchunk);
        if (splits > 1) {
            VAE_CHECK(q.slab, "f32 gemm: weight gradient needs %ld floats of split-K scratch", (long)splits * M * N);
            g.out = q.slab;
        }
    }
    const dim3 grid((unsigned)((M + TM - 1) / TM), (unsigned)((N + TN - 1) / TN), (unsigned)splits);
    VAE_CHECK(grid.y < 65536 && grid.z < 65536, "f32 gemm: grid too large");
    {
        ProfScope prof("gemm_f32_kernel", 2.0 * (double)M * (double)N * (double)K, 0.0, s);
#define GO(M, K) hipLaunchKernelGGL((gemm_f32_kernel<M, K>), grid, dim3(256), 0, s, g)
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
    if (splits > 1) {
        const long n = M * N;
        hipLaunchKernelGGL(fold_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, q.slab, splits, n, final_out);
        VAE_LAUNCH_CHECK("fold_f32_kernel");
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

__global__ __launch_bounds__(256) void bn_stats_f32_kernel(const float* __restrict__ y, int n_img, int C, int HW,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* running_mean, float* running_var, int train, float momentum,
                                                           float eps, float* coef, int cs) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    double mean, var;
    if (train) {
        double s = 0.0, q = 0.0;
        for (int n = 0; n < n_img; ++n) {
            const float* p = y + ((long)n * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 256) {
                const double v = (double)p[i];
                s += v;
                q = fma(v, v, q);
            }
        }
        s = block_sum_d(s, red);
        q = block_sum_d(q, red);
        const double cnt = (double)n_img * HW;
        mean = s / cnt;
        var = q / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        if (threadIdx.x == 0) {
            const double unb = var * (cnt / (cnt > 1.0 ? cnt - 1.0 : 1.0));
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        }
    } else {
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    }
    if (threadIdx.x == 0) {
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd;
        coef[c] = sc;
        coef[cs + c] = beta[c] - (float)mean * sc;
        coef[2 * cs + c] = (float)mean;
        coef[3 * cs + c] = rstd;
    }
}

int bn_stats(const float* y, int n_img, int C, int HW, const float* gamma, const float* beta, float* running_mean,
             float* running_var, int train, float momentum, float eps, float* coef, int coef_stride, hipStream_t s) {
    VAE_CHECK(y && gamma && beta && running_mean && running_var && coef, "f32 bn_stats: null pointer");
    VAE_CHECK(n_img > 0 && C > 0 && C <= coef_stride && HW > 0, "f32 bn_stats: bad geometry n=%d C=%d HW=%d", n_img, C, HW);
    hipLaunchKernelGGL(bn_stats_f32_kernel, dim3(C), dim3(256), 0, s, y, n_img, C, HW, gamma, beta, running_mean, running_var,
                       train, momentum, eps, coef, coef_stride);
    VAE_LAUNCH_CHECK("bn_stats_f32_kernel");
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
                                                                    const unsigned char* __restrict__ argmax, float* dgamma,
                                                                    float* dbeta, float* sums, int n_img, int C, int H, int W,
                                                                    int pool, const DropCfg d) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    const float sc = coef[c], sh = coef[cs + c], mean = coef[2 * cs + c], rstd = coef[3 * cs + c];
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < n_img; ++n)
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
        dbeta[c] = (float)s1;
        dgamma[c] = (float)s2;
        sums[c] = (float)s1;
        sums[C + c] = (float)s2;
    }
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
               float* dy, float* dgamma, float* dbeta, float* sums, int n_img, int C, int H, int W, int pool, const DropCfg& drop,
               hipStream_t s) {
    VAE_CHECK(g && y && coef && gamma && dy && dgamma && dbeta && sums, "f32 bn_act_bwd: null pointer");
    VAE_CHECK(n_img > 0 && C > 0 && H > 0 && W > 0, "f32 bn_act_bwd: bad geometry");
    VAE_CHECK(!pool || argmax, "f32 bn_act_bwd: the pooled form needs the forward's argmax bytes");
    hipLaunchKernelGGL(bn_act_bwd_reduce_f32_kernel, dim3(C), dim3(256), 0, s, g, y, coef, coef_stride, argmax, dgamma, dbeta, sums,
                       n_img, C, H, W, pool, drop);
    VAE_LAUNCH_CHECK("bn_act_bwd_reduce_f32_kernel");
    const long total = (long)n_img * C * H * W;
    hipLaunchKernelGGL(bn_act_bwd_apply_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, y, coef, coef_stride,
                       gamma, argmax, sums, dy, n_img, C, H, W, pool, drop);
    VAE_LAUNCH_CHECK("bn_act_bwd_apply_f32_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------- small pieces
__global__ __launch_bounds__(256) void chan_sum_f32_kernel(const float* __restrict__ t, int n_img, int C, int HW, float* out) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    double s = 0.0;
    const long per = (long)n_img * HW;
    for (long i = threadIdx.x; i < per; i += 256) {
        const long n = i / HW, r = i - n * HW;
        s += (double)t[(n * C + c) * HW + r];
    }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) out[c] = (float)s;
}
int chan_sum(const float* t, int n_img, int C, int HW, float* out, hipStream_t s) {
    VAE_CHECK(t && out && n_img > 0 && C > 0 && HW > 0, "f32 chan_sum: bad arguments");
    hipLaunchKernelGGL(chan_sum_f32_kernel, dim3(C), dim3(256), 0, s, t, n_img, C, HW, out);
    VAE_LAUNCH_CHECK("chan_sum_f32_kernel");
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
