// bf16 MFMA weight-gradient GEMM, "NT" form:  dW[m][tap][n] = sum_k P[rowP(k,tap)][m] * Q[rowQ(k,tap)][n]
//
// K is the pixel (or batch) index and is the STRIDED dimension of both operands (activations
// are NHWC: a row is one pixel, channels contiguous).  Tiles are staged row-wise into LDS
// (global_load_lds, 16 B per lane, rows stay coalesced) and the K-contiguous MFMA fragments are
// produced by the transposing LDS read ds_read_b64_tr_b16 -- no transposed copy of any
// activation ever exists in memory.
//
// Row gathers (what makes it a conv weight gradient):
//   G_PLAIN     rowP = rowQ = k                                      (linear layers)
//   G_CONV3     rowP = k, rowQ = pixel k shifted by the tap, zero outside the image   (3x3 conv)
//   G_CONVT     rowQ = k (input pixel), rowP = output pixel (2i+kh, 2j+kw)            (ConvTranspose k2s2)
//   G_CONV3_C8  rowP = k, Q is the 8-channel-padded network input: one 16-byte chunk per
//               tap, "tap" here selects taps 0-7 or 8-15 (taps >= 9 are zero)         (first conv)
//
// K is split over gridDim.z; every workgroup writes its fp32 partial tile to its own slab
// ([split][m][tap][n], plain 16-byte stores, deterministic) and vae_reduce_slabs folds the slabs
// into the gradient tensor in the reference layout.
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

enum { G_PLAIN = 0, G_CONV3 = 1, G_CONVT = 2, G_CONV3_C8 = 3 };

struct GemmNT {
    const bf16_t* P;
    const bf16_t* Q;
    float* slab;
    int M, N, Ktot, ldp, ldq;
    int H, W, wshift, hshift;   // pixel grid of k (shifts >= 0 when powers of two, else -1)
    int taps, kchunk;
    long slab_stride;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) short4v* l4ptr_t;

// Bank swizzle of the row-major [k][channel] LDS tiles.  One transposing read touches, per
// 32-lane half, 8 rows {b..b+3, b+8..b+11} x 32 contiguous bytes; unswizzled, rows 256 B (or
// 128 B) apart land on the same banks (8-way / 4-way conflict).  XOR-ing the 16-byte chunk
// index with an even number derived from the row spreads those 8 rows over 8 distinct 32-byte
// slots of the 256-byte bank line.  The image is written by global_load_lds (lane-linear), so the
// same XOR is applied to the SOURCE chunk each lane fetches.
template <int ROWB>
__device__ __forceinline__ int tr_swz(int row) {
    if (ROWB >= 256) return ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
    return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1;
}

// The two transposing reads of a fragment go through inline asm: while global->LDS copies are in flight the compiler
// puts s_waitcnt vmcnt(0) in front of every ds_read_b64_tr it issues itself (the copies are pending LDS writes to it, and
// it cannot tell the ring slot being filled from the one being read), which drains the whole ring at every K-step.
// The compiler does not count asm loads: frags_wait<N>() (at most N LDS reads still in flight) must precede the first
// use of a fragment.  `base` is an LDS byte address.
template <int ROWB>
__device__ __forceinline__ bf16x8 tr_frag(unsigned base, int k0, int col0, int lane) {
    // lanes 16g..16g+15 fetch k rows k0+8g .. k0+8g+7 of columns col0..col0+15, transposed:
    // lane (l&15) receives column col0+(l&15), elements = 8 consecutive k.
    const int g = lane >> 4, t = lane & 15;
    const int row = k0 + 8 * g + (t >> 2);
    const int off = (col0 + 4 * (t & 3)) * 2;            // byte offset inside the row
    const int chunk = off >> 4, within = off & 15;
    const unsigned a = base + row * ROWB + (((chunk ^ tr_swz<ROWB>(row)) << 4) | within);
    const unsigned b = base + (row + 4) * ROWB + (((chunk ^ tr_swz<ROWB>(row + 4)) << 4) | within);
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b) : "memory");
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
template <int N>
__device__ __forceinline__ void frags_wait(bf16x8& f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
}

// s_waitcnt vmcnt(groups * PER) for a wave-uniform runtime group count in [0, 3]
template <int PER>
__device__ __forceinline__ void nt_wait_groups(int groups) {
    if (groups <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (groups == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else if (groups == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
}

// NST-deep ring of K tiles: the copies of K-step ks + NST - 1 are issued while K-step ks computes.  These problems
// are short on arithmetic per byte (small M x N, long K), so what matters is how many tiles are in flight.
template <int BM, int BN, int MODE, int NST>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmNT p) {
    constexpr int BK = 64;
    constexpr int ROWP = BM * 2, ROWQ = BN * 2;          // LDS row bytes
    constexpr int P_BYTES = BK * ROWP, Q_BYTES = BK * ROWQ, STAGE = P_BYTES + Q_BYTES;
    constexpr int P_INSTR = P_BYTES / 1024 / 4, Q_INSTR = Q_BYTES / 1024 / 4;   // per wave
    constexpr int P_CH = ROWP / 16, Q_CH = ROWQ / 16;    // 16-byte chunks per row
    constexpr int TMw = BM / 2, TNw = BN / 2;            // wave tile (2 x 2 waves)
    constexpr int FM = TMw / 16, FN = TNw / 16;

    static_assert(NST >= 2 && NST <= 5 && (NST - 2) * (P_INSTR + Q_INSTR) <= 63, "ring depth");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: addresses and branches built on it stay scalar
    const int wn = wave & 1, wm = wave >> 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int tap = blockIdx.z % p.taps, split = blockIdx.z / p.taps;
    const int kbeg = split * p.kchunk;
    const int kend = min(kbeg + p.kchunk, p.Ktot);
    const bf16_t* zero = (const bf16_t*)g_zero_page;

    int dh = 0, dw = 0;
    if (MODE == G_CONV3) { dh = tap / 3 - 1; dw = tap % 3 - 1; }

    auto split_hw = [&](int k, int& h, int& w, int& img) {
        if (p.wshift >= 0 && p.hshift >= 0) {
            w = k & (p.W - 1);
            h = (k >> p.wshift) & (p.H - 1);
            img = k >> (p.wshift + p.hshift);
        } else {
            w = k % p.W;
            const int t = k / p.W;
            h = t % p.H;
            img = t / p.H;
        }
    };

    auto stage = [&](int buf, int k0) {
        unsigned char* pbase = smem + buf * STAGE;
        unsigned char* qbase = pbase + P_BYTES;
#pragma unroll
        for (int i = 0; i < P_INSTR; ++i) {
            const int t = wave + 4 * i;
            const int row = t * (64 / P_CH) + lane / P_CH, chunk = (lane % P_CH) ^ tr_swz<ROWP>(row);
            const int k = k0 + row;
            const bf16_t* src = zero;
            if (k < kend) {
                long r = k;
                if (MODE == G_CONVT) {
                    int h, w, img;
                    split_hw(k, h, w, img);
                    r = ((long)img * 2 * p.H + 2 * h + (tap >> 1)) * (2 * p.W) + 2 * w + (tap & 1);
                }
                src = p.P + r * p.ldp + m0 + chunk * 8;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(pbase + t * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < Q_INSTR; ++i) {
            const int t = wave + 4 * i;
            const int row = t * (64 / Q_CH) + lane / Q_CH, chunk = (lane % Q_CH) ^ tr_swz<ROWQ>(row);
            const int k = k0 + row;
            const bf16_t* src = zero;
            if (k < kend) {
                if (MODE == G_CONV3) {
                    int h, w, img;
                    split_hw(k, h, w, img);
                    if ((unsigned)(h + dh) < (unsigned)p.H && (unsigned)(w + dw) < (unsigned)p.W)
                        src = p.Q + (long)(k + dh * p.W + dw) * p.ldq + n0 + chunk * 8;
                } else if (MODE == G_CONV3_C8) {
                    const int tp = tap * 8 + chunk;        // one 16-byte chunk = one tap's 8 channels
                    int h, w, img;
                    split_hw(k, h, w, img);
                    const int eh = tp / 3 - 1, ew = tp % 3 - 1;
                    if (tp < 9 && (unsigned)(h + eh) < (unsigned)p.H && (unsigned)(w + ew) < (unsigned)p.W)
                        src = p.Q + (long)(k + eh * p.W + ew) * 8;
                } else {
                    src = p.Q + (long)k * p.ldq + n0 + chunk * 8;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(qbase + t * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;
#pragma unroll
    for (int q = 0; q < NST - 1; ++q)
        if (q < nk) stage(q, kbeg + q * BK);
    int slot = 0;
    for (int ks = 0; ks < nk; ++ks) {
        nt_wait_groups<P_INSTR + Q_INSTR>(min(NST - 2, nk - 1 - ks));      // K-step ks has landed
        barrier_keep_copies();            // not __syncthreads(): its fence would drain the younger copies too
        if (ks + NST - 1 < nk) stage(slot == 0 ? NST - 1 : slot - 1, kbeg + (ks + NST - 1) * BK);   // the slot read in K-step ks - 1
        const unsigned pbase = (unsigned)(size_t)(lptr_t)smem + (unsigned)(slot * STAGE);     // LDS byte addresses
        slot = slot + 1 == NST ? 0 : slot + 1;
        const unsigned qbase = pbase + P_BYTES;
        // the fragments of the second half of the K-step are requested before the first half is multiplied; LDS reads
        // complete in order, so a half is ready once no more reads are in flight than were issued after it
        bf16x8 qa[2][FN], pb[2][FM];
        auto load_half = [&](int kk) {
#pragma unroll
            for (int i = 0; i < FN; ++i) qa[kk][i] = tr_frag<ROWQ>(qbase, kk * 32, wn * TNw + i * 16, lane);
#pragma unroll
            for (int j = 0; j < FM; ++j) pb[kk][j] = tr_frag<ROWP>(pbase, kk * 32, wm * TMw + j * 16, lane);
        };
        load_half(0);
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            constexpr int NEWER = 2 * (FN + FM) < 15 ? 2 * (FN + FM) : 15;
            if (kk + 1 < BK / 32) {
                load_half(kk + 1);
#pragma unroll
                for (int i = 0; i < FN; ++i) frags_wait<NEWER>(qa[kk][i]);
            } else {
#pragma unroll
                for (int i = 0; i < FN; ++i) frags_wait<0>(qa[kk][i]);
            }
#pragma unroll
            for (int i = 0; i < FN; ++i)
#pragma unroll
                for (int j = 0; j < FM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[kk][i], pb[kk][j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // acc[i][j][r]: n = n0 + wn*TNw + i*16 + (lane>>4)*4 + r ; m = m0 + wm*TMw + j*16 + (lane&15)
    float* out = p.slab + (long)split * p.slab_stride;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
        const int m = m0 + wm * TMw + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int n = n0 + wn * TNw + i * 16 + (lane >> 4) * 4;
            if (m < p.M && n < p.N)
                *(float4*)(out + ((long)m * p.taps + tap) * p.N + n) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}

// out[e] (+)= sum_s slab[s][base + i0*s0 + i1*s1 + i2*s2 + i3*s3],  e = ((i0*D1 + i1)*D2 + i2)*D3 + i3
struct IndexMap {
    int d1, d2, d3;
    long s0, s1, s2, s3, base;
};

__device__ __forceinline__ long map_index(const IndexMap& mp, long e) {
    const int i3 = (int)(e % mp.d3);
    long t = e / mp.d3;
    const int i2 = (int)(t % mp.d2);
    t /= mp.d2;
    const int i1 = (int)(t % mp.d1);
    const long i0 = t / mp.d1;
    return mp.base + i0 * mp.s0 + i1 * mp.s1 + i2 * mp.s2 + i3 * mp.s3;
}

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, int nsplit,
                                                           long slab_stride, float* __restrict__ out,
                                                           long n, IndexMap mp, int accumulate) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const long src = map_index(mp, e);
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += slab[(long)k * slab_stride + src];
        out[e] = accumulate ? out[e] + s : s;
    }
}

// Same fold, but walking the SLAB in its own order (coalesced reads of every split) and
// scattering the few results: out[map(q)] (+)= sum_s slab[s][q], q over (D0,D1,D2,D3) = the slab's
// own dims; index tuples with i_d >= lim_d are padding and are skipped.
struct Limits {
    int l0, l1, l2, l3;
};
// SG split groups x 256/SG float4 columns per workgroup.  A job with many splits and few elements (the 64->64 conv:
// 256 slabs of 36 864 floats) is a chain of dependent loads per thread: with 4 groups that was 64 slabs per thread on
// 144 workgroups -- the long pole of its launch -- with 16 groups it is 16 slabs per thread (4 loads in flight) on 576.
template <int SG>
__device__ __forceinline__ void reduce_slabs_scatter_body(const float* __restrict__ slab, int nsplit, long slab_stride,
                                                          float* __restrict__ out, long n, const IndexMap& mp,
                                                          const Limits& lim, int accumulate, float4* red) {
    constexpr int CW = 256 / SG;
    const int ql = threadIdx.x % CW, sg = threadIdx.x / CW;
    const long n4 = n >> 2;
    const long st4 = slab_stride >> 2;
    for (long b = (long)blockIdx.x * CW; b < n4; b += (long)gridDim.x * CW) {
        const long q4 = b + ql;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        if (q4 < n4) {
            const float4* src = (const float4*)slab + q4;
            int k = sg;
            for (; k + 3 * SG < nsplit; k += 4 * SG) {
                const float4 u = src[(long)k * st4], v = src[(long)(k + SG) * st4], w = src[(long)(k + 2 * SG) * st4],
                             x = src[(long)(k + 3 * SG) * st4];
                a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
                a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
                a2.x += w.x; a2.y += w.y; a2.z += w.z; a2.w += w.w;
                a3.x += x.x; a3.y += x.y; a3.z += x.z; a3.w += x.w;
            }
            for (; k < nsplit; k += SG) {
                const float4 u = src[(long)k * st4];
                a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
            }
        }
        red[sg * CW + ql] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                        (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
        __syncthreads();
        // thread t < 4 * CW finishes element t & 3 of column t >> 2: the groups are added in a fixed order
        const int cq = threadIdx.x >> 2, el = threadIdx.x & 3;
        if (threadIdx.x < 4 * CW && b + cq < n4) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < SG; ++g) s += ((const float*)&red[g * CW + cq])[el];
            const long q = ((b + cq) << 2) + el;
            const int i3 = (int)(q % mp.d3);
            long t = q / mp.d3;
            const int i2 = (int)(t % mp.d2);
            t /= mp.d2;
            const int i1 = (int)(t % mp.d1);
            const long i0 = t / mp.d1;
            if (i0 < lim.l0 && i1 < lim.l1 && i2 < lim.l2 && i3 < lim.l3) {
                const long dst = mp.base + i0 * mp.s0 + i1 * mp.s1 + i2 * mp.s2 + i3 * mp.s3;
                out[dst] = accumulate ? out[dst] + s : s;
            }
        }
        __syncthreads();
    }
}

// Few slabs (<= 8) of many elements: no split groups, no LDS, no barrier.  A thread owns one float4 column, has every
// slab's value of it in flight at once, adds them as a fixed tree and scatters the four sums; the split-group form
// leaves most of its threads with one load (or none) per barrier-bounded round there.
__device__ __forceinline__ void reduce_slabs_direct_body(const float* __restrict__ slab, int nsplit, long slab_stride,
                                                         float* __restrict__ out, long n, const IndexMap& mp,
                                                         const Limits& lim, int accumulate) {
    const long n4 = n >> 2, st4 = slab_stride >> 2;
    for (long q4 = (long)blockIdx.x * 256 + threadIdx.x; q4 < n4; q4 += (long)gridDim.x * 256) {
        const float4* src = (const float4*)slab + q4;
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = k < nsplit ? src[(long)k * st4] : make_float4(0.f, 0.f, 0.f, 0.f);
        float s[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* f = (const float*)v + e;
            s[e] = ((f[0] + f[4]) + (f[8] + f[12])) + ((f[16] + f[20]) + (f[24] + f[28]));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned q = (unsigned)(q4 << 2) + e;       // n < 2^31 (checked by the launchers)
            const unsigned d3 = mp.d3, d2 = mp.d2, d1 = mp.d1;
            const unsigned i3 = q % d3;
            unsigned t = q / d3;
            const unsigned i2 = t % d2;
            t /= d2;
            const unsigned i1 = t % d1, i0 = t / d1;
            if ((int)i0 < lim.l0 && (int)i1 < lim.l1 && (int)i2 < lim.l2 && (int)i3 < lim.l3) {
                const long dst = mp.base + i0 * mp.s0 + i1 * mp.s1 + i2 * mp.s2 + i3 * mp.s3;
                out[dst] = accumulate ? out[dst] + s[e] : s[e];
            }
        }
    }
}

int g_reduce_direct_mode = 1;
// 0: four split groups x 64 columns, 1: sixteen x 16, 2: the direct form
static inline int reduce_form(int nsplit) { return nsplit >= 64 ? 1 : (g_reduce_direct_mode && nsplit <= 8) ? 2 : 0; }
static inline long reduce_cols(int form) { return form == 1 ? 16 : form == 2 ? 256 : 64; }

__global__ __launch_bounds__(256) void reduce_slabs_scatter_kernel(const float* __restrict__ slab, int nsplit,
                                                                   long slab_stride, float* __restrict__ out, long n,
                                                                   IndexMap mp, Limits lim, int accumulate, int sg16) {
    __shared__ float4 red[256];
    if (sg16 == 2) reduce_slabs_direct_body(slab, nsplit, slab_stride, out, n, mp, lim, accumulate);
    else if (sg16) reduce_slabs_scatter_body<16>(slab, nsplit, slab_stride, out, n, mp, lim, accumulate, red);
    else reduce_slabs_scatter_body<4>(slab, nsplit, slab_stride, out, n, mp, lim, accumulate, red);
}

// several folds in one launch: blockIdx.y selects the job
struct ReduceJob {
    const float* slab;
    float* out;
    long slab_stride, n;
    IndexMap mp;
    Limits lim;
    int nsplit, accumulate, sg16;      // sg16: the form (see reduce_form)
};
struct ReduceJobs {
    ReduceJob j[VAE_MAX_REDUCE_JOBS];
};
__global__ __launch_bounds__(256) void multi_reduce_slabs_scatter_kernel(const ReduceJobs jobs) {
    __shared__ float4 red[256];
    const ReduceJob& jb = jobs.j[blockIdx.y];
    if ((long)blockIdx.x * (jb.sg16 == 2 ? 256 : jb.sg16 ? 16 : 64) >= (jb.n >> 2)) return;      // whole workgroup: no barrier is skipped by part of it
    if (jb.sg16 == 2) reduce_slabs_direct_body(jb.slab, jb.nsplit, jb.slab_stride, jb.out, jb.n, jb.mp, jb.lim, jb.accumulate);
    else if (jb.sg16) reduce_slabs_scatter_body<16>(jb.slab, jb.nsplit, jb.slab_stride, jb.out, jb.n, jb.mp, jb.lim, jb.accumulate, red);
    else reduce_slabs_scatter_body<4>(jb.slab, jb.nsplit, jb.slab_stride, jb.out, jb.n, jb.mp, jb.lim, jb.accumulate, red);
}

// Every weight tensor of the model repacked in ONE launch: blockIdx.y selects the job.  A job is walked in the
// DESTINATION's order (its four dims sorted by destination stride on the host), so the packed operand is written
// in runs -- 16 bytes per lane when the innermost run is a multiple of 8 bf16 -- and the fp32 source is gathered
// (strided reads cost a fraction of what scattered 2-byte stores do).
struct PackJob {
    const float* src;
    void* dst;
    int n;               // elements
    int d1, d2, d3;      // sizes of the three inner dims in destination order
    int ss[4];           // source strides (elements) of the four dims in destination order
    long ds[4];          // destination strides, ds[3] == 1 when vec8
    long base;
    int f32;             // 1: fp32 permuting copy, 0: bf16 pack
    int vec8;            // innermost destination run is contiguous and a multiple of 8
    // tiled form (a transposition: the source's fastest dim is not the destination's): 64 x 32 tiles through LDS,
    // read along the source's fastest dim `si`, written along the destination's fastest dim `di`
    int tiled, si, di, r0, r1;            // tiled: 0 gather, 1 tiles over (di, si), 2 tiles over (dim 0, inner run); r0, r1: the two other dims
    int dim[4];                           // sizes in SOURCE order
    long dst_stride[4];                   // destination strides in source order (dst_stride[di] == 1)
    int tiles_d, tiles_s;
};
struct PackJobs {
    PackJob j[VAE_MAX_PACK_JOBS];
};
// A job whose source and destination run along different dims (the nn.Linear weights: NCHW-flatten columns -> NHWC
// order, and the transposed copies for the input-gradient GEMMs).  Gathering 4-byte elements a row apart costs a
// memory transaction per element; here a 64 (destination-fastest) x 32 (source-fastest) tile is read in 128-byte
// runs, turned in LDS and written as 16 bytes (8 bf16) per lane in 128-byte runs.
__device__ __forceinline__ void pack_tiled(const PackJob& jb, float (*tile)[33]) {
    const long ntile = (long)jb.dim[jb.r0] * jb.dim[jb.r1] * jb.tiles_d * jb.tiles_s;
    const long sstride[4] = {(long)jb.dim[1] * jb.dim[2] * jb.dim[3], (long)jb.dim[2] * jb.dim[3], jb.dim[3], 1};
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        long q = t;
        const int ts = (int)(q % jb.tiles_s); q /= jb.tiles_s;
        const int td = (int)(q % jb.tiles_d); q /= jb.tiles_d;
        const int i1 = (int)(q % jb.dim[jb.r1]);
        const int i0 = (int)(q / jb.dim[jb.r1]);
        const long sbase = i0 * sstride[jb.r0] + i1 * sstride[jb.r1];
        const long dbase = jb.base + i0 * jb.dst_stride[jb.r0] + i1 * jb.dst_stride[jb.r1];
        const int d0 = td * 64, s0 = ts * 32;
        // read: lane -> source-fastest index (32 consecutive floats per row), 8 destination rows per pass
        {
            const int sl = threadIdx.x & 31, dl = threadIdx.x >> 5;
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int d = d0 + dl + 8 * ps, sx = s0 + sl;
                float v = 0.f;
                if (d < jb.dim[jb.di] && sx < jb.dim[jb.si]) v = jb.src[sbase + (long)d * sstride[jb.di] + sx];
                tile[dl + 8 * ps][sl] = v;
            }
        }
        __syncthreads();
        // write: lane -> 8 consecutive destination elements of one source-fastest index
        {
            const int d8 = (threadIdx.x & 7) * 8, sl = threadIdx.x >> 3;
            const int sx = s0 + sl;
            if (sx < jb.dim[jb.si] && d0 + d8 < jb.dim[jb.di]) {       // dim[di] % 8 == 0 (checked on the host)
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tile[d8 + k][sl];
                const long d = dbase + (long)sx * jb.dst_stride[jb.si] + d0 + d8;
                *(uint4*)((bf16_t*)jb.dst + d) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
            }
        }
        __syncthreads();
    }
}

// The same turn for a job whose FIRST source dim is the destination's fastest while the source's own fastest dim is
// short (the conv weights for the input-gradient GEMMs: [co][ci][tap] -> [ci][8 - tap][co]; the ConvTranspose weights
// [ci][co][tap] -> [tap][co][ci]): the three inner source dims are read as one contiguous run of Q = d1 * d2 * d3
// floats per row, and a tile's 32 positions of that run are taken apart again for the destination offset.
__device__ __forceinline__ void pack_tiled_flat(const PackJob& jb, float (*tile)[33]) {
    const int D0 = jb.dim[0], Q = jb.dim[1] * jb.dim[2] * jb.dim[3];
    const int ntile = jb.tiles_d * jb.tiles_s;
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int d0 = (t / jb.tiles_s) * 64, s0 = (t % jb.tiles_s) * 32;
        {
            const int sl = threadIdx.x & 31, dl = threadIdx.x >> 5;
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int d = d0 + dl + 8 * ps, sx = s0 + sl;
                tile[dl + 8 * ps][sl] = (d < D0 && sx < Q) ? jb.src[(long)d * Q + sx] : 0.f;
            }
        }
        __syncthreads();
        {
            const int d8 = (threadIdx.x & 7) * 8, sl = threadIdx.x >> 3;
            const int sx = s0 + sl;
            if (sx < Q && d0 + d8 < D0) {                              // dim[0] % 8 == 0 (checked on the host)
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tile[d8 + k][sl];
                const int i3 = sx % jb.dim[3], r = sx / jb.dim[3];
                const int i2 = r % jb.dim[2], i1 = r / jb.dim[2];
                const long d = jb.base + i1 * jb.dst_stride[1] + i2 * jb.dst_stride[2] + i3 * jb.dst_stride[3] + d0 + d8;
                *(uint4*)((bf16_t*)jb.dst + d) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void multi_pack_kernel(const PackJobs jobs) {
    __shared__ float tile[64][33];
    const PackJob& jb = jobs.j[blockIdx.y];
    if (jb.tiled == 1) { pack_tiled(jb, tile); return; }
    if (jb.tiled == 2) { pack_tiled_flat(jb, tile); return; }
    const int step = jb.vec8 ? 8 : 1;
    const int items = jb.n / step;
    // Workgroups go round-robin over the 8 XCDs, each with its own L2, and neighbouring destination runs gather from
    // the same source lines: give every XCD one contiguous eighth of the job (gridDim.x is a multiple of 8, or < 8)
    int begin = blockIdx.x * blockDim.x, end = items, stride = gridDim.x * blockDim.x;
    if ((gridDim.x & 7) == 0) {
        const int per = (items + 7) >> 3, xcd = blockIdx.x & 7;
        begin = xcd * per + (blockIdx.x >> 3) * blockDim.x;
        end = min(items, (xcd + 1) * per);
        stride = (gridDim.x >> 3) * blockDim.x;
    }
    for (int it = begin + threadIdx.x; it < end; it += stride) {
        const unsigned e = (unsigned)it * step;
        const unsigned i3 = e % (unsigned)jb.d3;
        unsigned t = e / (unsigned)jb.d3;
        const unsigned i2 = t % (unsigned)jb.d2;
        t /= (unsigned)jb.d2;
        const unsigned i1 = t % (unsigned)jb.d1, i0 = t / (unsigned)jb.d1;
        const float* s = jb.src + ((long)i0 * jb.ss[0] + (long)i1 * jb.ss[1] + (long)i2 * jb.ss[2] + (long)i3 * jb.ss[3]);
        const long d = jb.base + i0 * jb.ds[0] + i1 * jb.ds[1] + i2 * jb.ds[2] + i3 * jb.ds[3];
        if (jb.vec8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = s[(long)k * jb.ss[3]];
            *(uint4*)((bf16_t*)jb.dst + d) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        } else if (jb.f32) {
            ((float*)jb.dst)[d] = *s;
        } else {
            ((bf16_t*)jb.dst)[d] = f2bf(*s);
        }
    }
}

__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                        long n, IndexMap mp) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        dst[map_index(mp, e)] = f2bf(src[e]);
}

__global__ __launch_bounds__(256) void permute_f32_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          long n, IndexMap mp) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        dst[map_index(mp, e)] = src[e];
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        dst[e] = f2bf(src[e]);
}

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

template <int MODE>
static int launch_nt(const GemmNT& p, int nsplit, hipStream_t s) {
    const bool m128 = p.M % 128 == 0, n128 = p.N % 128 == 0;
    dim3 block(256);
    // algorithmic work: every k row of P and Q read once, the slabs written once
    const double ntap = MODE == G_CONV3_C8 ? 27.0 / 64.0 : (double)p.taps;   // 3 real channels x 9 taps of the 2 x 64 columns
    const double flops = 2.0 * p.M * p.N * ntap * p.Ktot;
    const double bytes = 2.0 * p.Ktot * ((double)p.M * (MODE == G_CONVT ? 4 : 1) + (MODE == G_CONV3_C8 ? 8 : p.N)) +
                         4.0 * nsplit * p.taps * p.M * p.N;
#define GO(BM_, BN_)                                                                          \
    {                                                                                         \
        dim3 grid(p.N / BN_, p.M / BM_, p.taps * nsplit);                                     \
        static char name[64] = "";                                                            \
        if (!name[0]) snprintf(name, sizeof(name), "gemm_nt_kernel<%d,%d,%d>", BM_, BN_, MODE); \
        ProfScope prof(name, flops, bytes, s);                                                \
        /* ring depth: what lets two workgroups share a CU's LDS; four when the grid gives a CU one workgroup anyway */ \
        constexpr int STG = (BM_ + BN_) * 128;                                                \
        constexpr int NST2 = STG <= 16384 ? 4 : STG <= 24576 ? 3 : 2;                         \
        constexpr int NST1 = 4;                                                               \
        if (NST1 != NST2 && (long)grid.x * grid.y * grid.z <= 256)                            \
            hipLaunchKernelGGL((gemm_nt_kernel<BM_, BN_, MODE, NST1>), grid, block, 0, s, p); \
        else                                                                                  \
            hipLaunchKernelGGL((gemm_nt_kernel<BM_, BN_, MODE, NST2>), grid, block, 0, s, p); \
    }
    if (m128 && n128) GO(128, 128)
    else if (m128) GO(128, 64)
    else if (n128) GO(64, 128)
    else GO(64, 64)
#undef GO
    VAE_LAUNCH_CHECK("gemm_nt_kernel");
    return 0;
}

}  // namespace vae

using namespace vae;

extern "C" int vae_wgrad_ksplit(int M, int N, int taps, long Ktot) {
    // aim at ~512 workgroups (2 per CU) with at least 256 k rows (four K-steps) each.  A K-step of these
    // small tiles costs about a memory round trip (~1.5 us) whatever the ring depth, a split costs one slab
    // of the weight tensor written and read back (~0.1-0.3 us): short K ranges on many CUs win
    const long tm = M % 128 == 0 ? M / 128 : M / 64, tn = N % 128 == 0 ? N / 128 : N / 64;
    const long tiles = tm * tn * taps;
    long s = (512 + tiles - 1) / tiles;
    const long maxs = (Ktot + 255) / 256;
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    // with the ring of K tiles working, 2048 k rows per workgroup on 256 workgroups cost the same as 1024 rows on 512
    // (ConvTranspose 64->64 at 16x16, batch 512: 22.0 vs 22.5 us) and leave half the slabs to fold
    if (tiles * s >= 512 && Ktot / s < 2048) s = (s + 1) / 2;
    // normalise: chunks are multiples of 64 k rows and no split may be empty
    const long chunk = ((Ktot + s - 1) / s + 63) / 64 * 64;
    s = (Ktot + chunk - 1) / chunk;
    return (int)s;
}

extern "C" int vae_wgrad_gemm(const void* P, const void* Q, float* slab, int M, int N, long Ktot, int ldp,
                              int ldq, int mode, int H, int W, int nsplit, void* stream) {
    VAE_CHECK(P && Q && slab, "vae_wgrad_gemm: null pointer");
    VAE_CHECK(M > 0 && N > 0 && Ktot > 0 && Ktot < (1L << 31) && nsplit > 0, "vae_wgrad_gemm: bad sizes M=%d N=%d K=%ld S=%d", M, N, Ktot, nsplit);
    VAE_CHECK(M % 64 == 0 && N % 64 == 0, "vae_wgrad_gemm: M=%d and N=%d must be multiples of 64", M, N);
    VAE_CHECK((((uintptr_t)P | (uintptr_t)Q | (uintptr_t)slab) & 15) == 0 && ldp % 8 == 0 && ldq % 8 == 0,
              "vae_wgrad_gemm: operands must be 16-byte aligned with row strides a multiple of 8");
    GemmNT p = {};
    p.P = (const bf16_t*)P; p.Q = (const bf16_t*)Q; p.slab = slab;
    p.M = M; p.N = N; p.Ktot = (int)Ktot; p.ldp = ldp; p.ldq = ldq;
    p.H = H; p.W = W; p.wshift = ilog2_exact(W); p.hshift = ilog2_exact(H);
    const long chunk = ((Ktot + nsplit - 1) / nsplit + 63) / 64 * 64;
    p.kchunk = (int)chunk;
    VAE_CHECK((long)(nsplit - 1) * chunk < Ktot, "vae_wgrad_gemm: nsplit=%d leaves empty splits for K=%ld", nsplit, Ktot);
    hipStream_t s = (hipStream_t)stream;
    switch (mode) {
        case VAE_WGRAD_PLAIN: p.taps = 1; p.slab_stride = (long)M * N; return launch_nt<G_PLAIN>(p, nsplit, s);
        case VAE_WGRAD_CONV3:
            VAE_CHECK(H > 0 && W > 0 && Ktot % ((long)H * W) == 0, "vae_wgrad_gemm: K=%ld is not a whole number of %dx%d images", Ktot, H, W);
            p.taps = 9; p.slab_stride = 9L * M * N; return launch_nt<G_CONV3>(p, nsplit, s);
        case VAE_WGRAD_CONVT:
            VAE_CHECK(H > 0 && W > 0 && Ktot % ((long)H * W) == 0, "vae_wgrad_gemm: K=%ld is not a whole number of %dx%d images", Ktot, H, W);
            p.taps = 4; p.slab_stride = 4L * M * N; return launch_nt<G_CONVT>(p, nsplit, s);
        case VAE_WGRAD_CONV3_C8:
            VAE_CHECK(N == 64 && H > 0 && W > 0 && Ktot % ((long)H * W) == 0, "vae_wgrad_gemm: the 8-channel input mode needs N == 64 (8 taps x 8 channels)");
            p.taps = 2; p.slab_stride = 2L * M * N; return launch_nt<G_CONV3_C8>(p, nsplit, s);
        default: VAE_CHECK(false, "vae_wgrad_gemm: unknown mode %d", mode);
    }
    return 0;
}

static int make_map(IndexMap& mp, const long* dims, const long* strides, long base, const char* who) {
    VAE_CHECK(dims && strides, "%s: null dims/strides", who);
    VAE_CHECK(dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && dims[3] > 0 && dims[1] < (1L << 31) &&
                  dims[2] < (1L << 31) && dims[3] < (1L << 31), "%s: bad dims", who);
    mp.d1 = (int)dims[1]; mp.d2 = (int)dims[2]; mp.d3 = (int)dims[3];
    mp.s0 = strides[0]; mp.s1 = strides[1]; mp.s2 = strides[2]; mp.s3 = strides[3];
    mp.base = base;
    return 0;
}

extern "C" int vae_reduce_slabs(const float* slab, int nsplit, long slab_stride, float* out, const long* dims4,
                                const long* strides4, long base, int accumulate, void* stream) {
    VAE_CHECK(slab && out && nsplit > 0, "vae_reduce_slabs: bad arguments");
    IndexMap mp;
    if (int e = make_map(mp, dims4, strides4, base, "vae_reduce_slabs")) return e;
    const long n = dims4[0] * dims4[1] * dims4[2] * dims4[3];
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    ProfScope prof("reduce_slabs_kernel", 0.0, 4.0 * n * (nsplit + 1), (hipStream_t)stream);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab, nsplit,
                       slab_stride, out, n, mp, accumulate);
    VAE_LAUNCH_CHECK("reduce_slabs_kernel");
    return 0;
}

extern "C" int vae_pack_bf16(const float* src, void* dst, const long* dims4, const long* strides4, long base,
                             void* stream) {
    VAE_CHECK(src && dst, "vae_pack_bf16: null pointer");
    IndexMap mp;
    if (int e = make_map(mp, dims4, strides4, base, "vae_pack_bf16")) return e;
    const long n = dims4[0] * dims4[1] * dims4[2] * dims4[3];
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    ProfScope prof("pack_bf16_kernel", 0.0, 6.0 * n, (hipStream_t)stream);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n, mp);
    VAE_LAUNCH_CHECK("pack_bf16_kernel");
    return 0;
}

extern "C" int vae_permute_f32(const float* src, float* dst, const long* dims4, const long* strides4, long base,
                               void* stream) {
    VAE_CHECK(src && dst, "vae_permute_f32: null pointer");
    IndexMap mp;
    if (int e = make_map(mp, dims4, strides4, base, "vae_permute_f32")) return e;
    const long n = dims4[0] * dims4[1] * dims4[2] * dims4[3];
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(permute_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, n, mp);
    VAE_LAUNCH_CHECK("permute_f32_kernel");
    return 0;
}

extern "C" int vae_cast_bf16(const float* src, void* dst, long n, void* stream) {
    VAE_CHECK(src && dst && n > 0, "vae_cast_bf16: bad arguments");
    const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    VAE_LAUNCH_CHECK("cast_bf16_kernel");
    return 0;
}

extern "C" int vae_reduce_slabs_scatter(const float* slab, int nsplit, long slab_stride, long slab_base, float* out,
                                        const long* dims4, const long* strides4, const long* limits4, long base,
                                        int accumulate, void* stream) {
    VAE_CHECK(slab && out && nsplit > 0, "vae_reduce_slabs_scatter: bad arguments");
    IndexMap mp;
    if (int e = make_map(mp, dims4, strides4, base, "vae_reduce_slabs_scatter")) return e;
    Limits lim = {(int)dims4[0], (int)dims4[1], (int)dims4[2], (int)dims4[3]};
    if (limits4) lim = {(int)limits4[0], (int)limits4[1], (int)limits4[2], (int)limits4[3]};
    const long n = dims4[0] * dims4[1] * dims4[2] * dims4[3];
    VAE_CHECK(n % 4 == 0 && slab_stride % 4 == 0 && slab_base % 4 == 0 && ((uintptr_t)slab & 15) == 0,
              "vae_reduce_slabs_scatter: element count, slab stride and base must be multiples of 4 (16-byte rows)");
    VAE_CHECK(n < (1L << 31), "vae_reduce_slabs_scatter: more than 2^31 elements");
    const int sg16 = reduce_form(nsplit);
    const long cw = reduce_cols(sg16), blocks = (n / 4 + cw - 1) / cw;
    const int grid = (int)(blocks > 2048 ? 2048 : blocks);
    ProfScope prof("reduce_slabs_scatter_kernel", 0.0, 4.0 * n * (nsplit + 1), (hipStream_t)stream);
    hipLaunchKernelGGL(reduce_slabs_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab + slab_base,
                       nsplit, slab_stride, out, n, mp, lim, accumulate, sg16);
    VAE_LAUNCH_CHECK("reduce_slabs_scatter_kernel");
    return 0;
}

extern "C" int vae_multi_reduce_slabs_scatter(const vae_reduce_job* jobs, int njobs, void* stream) {
    VAE_CHECK(jobs && njobs > 0 && njobs <= VAE_MAX_REDUCE_JOBS, "vae_multi_reduce_slabs_scatter: njobs=%d out of range (1..%d)",
              njobs, VAE_MAX_REDUCE_JOBS);
    ReduceJobs rj;
    long max_blocks = 1;
    double bytes = 0;
    for (int i = 0; i < njobs; ++i) {
        const vae_reduce_job& q = jobs[i];
        VAE_CHECK(q.slab && q.out && q.nsplit > 0, "vae_multi_reduce_slabs_scatter: job %d has bad arguments", i);
        if (int e = make_map(rj.j[i].mp, q.dims, q.strides, q.base, "vae_multi_reduce_slabs_scatter")) return e;
        rj.j[i].lim = {(int)q.limits[0], (int)q.limits[1], (int)q.limits[2], (int)q.limits[3]};
        const long n = q.dims[0] * q.dims[1] * q.dims[2] * q.dims[3];
        VAE_CHECK(n % 4 == 0 && q.slab_stride % 4 == 0 && q.slab_base % 4 == 0 && ((uintptr_t)q.slab & 15) == 0,
                  "vae_multi_reduce_slabs_scatter: job %d: element count, slab stride and base must be multiples of 4", i);
        rj.j[i].slab = q.slab + q.slab_base; rj.j[i].out = q.out; rj.j[i].slab_stride = q.slab_stride; rj.j[i].n = n;
        rj.j[i].nsplit = q.nsplit; rj.j[i].accumulate = q.accumulate;
        VAE_CHECK(n < (1L << 31), "vae_multi_reduce_slabs_scatter: job %d has more than 2^31 elements", i);
        rj.j[i].sg16 = reduce_form(q.nsplit);
        const long cw = reduce_cols(rj.j[i].sg16);
        const long blocks = (n / 4 + cw - 1) / cw;
        if (blocks > max_blocks) max_blocks = blocks;
        bytes += 4.0 * n * (q.nsplit + 1);
    }
    if (max_blocks > 2048) max_blocks = 2048;
    // Workgroups are dispatched job by job: the jobs with the most slabs -- whose threads walk the longest chains of
    // dependent round trips -- go first, so that their tails run beside the wide, short jobs instead of after them.
    // (The jobs are independent: the order changes no result.)
    for (int i = 1; i < njobs; ++i)
        for (int k = i; k > 0 && rj.j[k].nsplit > rj.j[k - 1].nsplit; --k) { const ReduceJob t = rj.j[k]; rj.j[k] = rj.j[k - 1]; rj.j[k - 1] = t; }
    ProfScope prof("reduce_slabs_scatter_kernel", 0.0, bytes, (hipStream_t)stream);
    hipLaunchKernelGGL(multi_reduce_slabs_scatter_kernel, dim3((int)max_blocks, njobs), dim3(256), 0, (hipStream_t)stream, rj);
    VAE_LAUNCH_CHECK("multi_reduce_slabs_scatter_kernel");
    return 0;
}

// jobs: njobs x {src, dst, n, dims[3] (D1..D3), strides[4], base, f32}; one launch for all of them
extern "C" int vae_multi_pack(const vae_pack_job* jobs, int njobs, void* stream) {
    VAE_CHECK(jobs && njobs > 0 && njobs <= VAE_MAX_PACK_JOBS, "vae_multi_pack: njobs=%d out of range (1..%d)", njobs,
              VAE_MAX_PACK_JOBS);
    PackJobs pj;
    double bytes = 0;
    int max_items = 1;
    for (int i = 0; i < njobs; ++i) {
        VAE_CHECK(jobs[i].src && jobs[i].dst, "vae_multi_pack: job %d has a null pointer", i);
        const long* dims = jobs[i].dims;
        VAE_CHECK(dims[0] > 0 && dims[1] > 0 && dims[2] > 0 && dims[3] > 0, "vae_multi_pack: job %d has bad dims", i);
        const long n = dims[0] * dims[1] * dims[2] * dims[3];
        VAE_CHECK(n < (1L << 31), "vae_multi_pack: job %d has 2^31 or more elements", i);
        // order the dims by destination stride, largest first (size-1 dims first of all: they index nothing)
        const long sstride[4] = {dims[1] * dims[2] * dims[3], dims[2] * dims[3], dims[3], 1};
        int ord[4] = {0, 1, 2, 3};
        auto key = [&](int k) { return dims[k] == 1 ? (1L << 62) : (jobs[i].strides[k] < 0 ? -jobs[i].strides[k] : jobs[i].strides[k]); };
        for (int a = 0; a < 4; ++a)
            for (int b = a + 1; b < 4; ++b)
                if (key(ord[b]) > key(ord[a])) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
        PackJob& q = pj.j[i];
        q.src = jobs[i].src; q.dst = jobs[i].dst; q.n = (int)n; q.f32 = jobs[i].f32; q.base = jobs[i].base;
        q.d1 = (int)dims[ord[1]]; q.d2 = (int)dims[ord[2]]; q.d3 = (int)dims[ord[3]];
        for (int k = 0; k < 4; ++k) { q.ss[k] = (int)sstride[ord[k]]; q.ds[k] = jobs[i].strides[ord[k]]; }
        bool aligned = jobs[i].base % 8 == 0 && ((uintptr_t)jobs[i].dst & 15) == 0;
        for (int k = 0; k < 3; ++k) aligned = aligned && (dims[ord[k]] == 1 || q.ds[k] % 8 == 0);
        q.vec8 = !jobs[i].f32 && q.ds[3] == 1 && q.d3 % 8 == 0 && aligned;
        // a transposition of two large dims goes through LDS tiles: source-fastest dim si (the last dim of size > 1),
        // destination-fastest dim di (stride 1)
        q.tiled = 0;
        int si = 3;
        while (si > 0 && dims[si] == 1) --si;
        const int di = ord[3];
        if (q.vec8 && si != di && dims[si] >= 16 && dims[di] >= 64 && dims[di] % 8 == 0) {
            q.tiled = 1; q.si = si; q.di = di;
            int rest[2], nr = 0;
            for (int k = 0; k < 4; ++k) if (k != si && k != di) rest[nr++] = k;
            q.r0 = rest[0]; q.r1 = rest[1];
            for (int k = 0; k < 4; ++k) { q.dim[k] = (int)dims[k]; q.dst_stride[k] = jobs[i].strides[k]; }
            q.tiles_d = (int)((dims[di] + 63) / 64); q.tiles_s = (int)((dims[si] + 31) / 32);
        } else if (q.vec8 && di == 0 && si != 0 && dims[0] >= 64 && dims[0] % 8 == 0 && sstride[0] >= 32) {
            // the inner source dims as one run (the source's fastest dim alone is too short for a tile)
            q.tiled = 2; q.si = si; q.di = 0; q.r0 = q.r1 = 0;
            for (int k = 0; k < 4; ++k) { q.dim[k] = (int)dims[k]; q.dst_stride[k] = jobs[i].strides[k]; }
            q.tiles_d = (int)((dims[0] + 63) / 64); q.tiles_s = (int)((sstride[0] + 31) / 32);
        }
        const int items = q.tiled ? (int)(n / 8) : (int)(n / (q.vec8 ? 8 : 1));      // a tile holds 2048 elements = 256 items
        if (items > max_items) max_items = items;
        bytes += n * (jobs[i].f32 ? 8.0 : 6.0);
    }
    // the largest jobs are dispatched first (the jobs are independent: their order changes no result)
    for (int i = 1; i < njobs; ++i)
        for (int k = i; k > 0 && pj.j[k].n > pj.j[k - 1].n; --k) { const PackJob t = pj.j[k]; pj.j[k] = pj.j[k - 1]; pj.j[k - 1] = t; }
    int gx = (max_items + 255) / 256;
    if (gx > 1024) gx = 1024;
    if (gx > 8) gx &= ~7;          // whole rounds of the 8 XCDs (the kernel then keeps each XCD on its own part of a job)
    ProfScope prof("multi_pack_kernel", 0.0, bytes, (hipStream_t)stream);
    hipLaunchKernelGGL(multi_pack_kernel, dim3(gx, njobs), dim3(256), 0, (hipStream_t)stream, pj);
    VAE_LAUNCH_CHECK("multi_pack_kernel");
    return 0;
}
