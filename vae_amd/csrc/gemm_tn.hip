// bf16 MFMA implicit-GEMM, "TN" form:  out[m][n] = sum_k A[m][k] * B[n][k]  (fp32 accumulate).
//
// Both operands are K-contiguous in memory, so every LDS row is a 128-byte (BK=64) or
// 64-byte (BK=32) run of one operand row and MFMA fragments are plain ds_read_b128.
// The A side (activations, NHWC) is addressed through a per-row gather, which is what makes
// the GEMM "implicit":
//   A_PLAIN        row m = A + m*lda                                   (linear, im2col'd conv1a, convT fwd)
//   A_CONV3        row m = pixel (img,h,w); K = (tap, c): the tap's neighbour pixel or the
//                  zero page when it falls in the padding              (3x3 s1 p1 conv fwd and dgrad)
//   A_CONVT_DGRAD  row m = input pixel (img,i,j); K = (tap, c): output pixel (2i+kh, 2j+kw)
//                                                                      (ConvTranspose2d k2 s2 dgrad)
//   A_CONV3_C8     the network input padded to 8 channels (16 B per pixel): every 16-byte chunk
//                  of a row is ONE tap's pixel; K = 16 taps x 8 channels, taps >= 9 are zero
//                                                                      (first conv, Cin <= 8)
// Tiles are staged global -> LDS with buffer_load ... lds (16 B per lane, no VGPR round trip;
// out-of-range offsets deliver the zeros of the padding); the LDS image is lane-linear, the XOR
// bank swizzle is applied to the per-lane SOURCE chunk and again on the fragment read.  The
// stages form a ring of 2-4 K-steps, one barrier per K-step.
//
// MFMA orientation: weights are the MFMA "A" operand (rows = output channels) and activations
// the "B" operand (columns = pixels), so a lane ends up holding 4 CONSECUTIVE channels of one
// pixel per accumulator -> 8-byte (bf16) / 16-byte (fp32) NHWC stores, and per-channel batch
// statistics reduce over the 16 lanes that share (lane >> 4).
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

// conv3_patch_tn.hip
bool conv3_patch_tn_ok(int n_img, int H, int W, int Cin, int Cout);
int conv3_patch_tn_rows(long M, int H, int W, int Cin, int Cout);
int conv3_patch_tn_launch(const void* x, const void* w_packed, const float* bias, void* y, float* stat_sum, float* stat_sq,
                          int n_img, int H, int W, int Cin, int Cout, double flops, double bytes, hipStream_t s);

// stream_gemm.hip
bool stream_c8_ok(int n_img, int H, int W, int Cout);
int stream_c8_rows(int n_img, int H, int W);
int stream_c8_launch(const void* x8, const void* w_packed, const float* bias, void* y, float* stat_sum, float* stat_sq,
                     const float* scale, const float* shift, void* relu_bits, int n_img, int H, int W, int Cout, double flops,
                     double bytes, hipStream_t s);
bool stream_convt_fwd_ok(int n_img, int H, int W, int Cin, int Cout);
int stream_convt_fwd_rows(int n_img, int H, int W, int Cin, int Cout);
int stream_convt_fwd_fold(int Cin, int Cout);
int stream_convt_fwd_launch(const void* x, const void* w_packed, const float* bias, void* y, float* stat_sum, float* stat_sq,
                            int n_img, int H, int W, int Cin, int Cout, double flops, double bytes, hipStream_t s);

enum { A_PLAIN = 0, A_CONV3 = 1, A_CONVT_DGRAD = 2, A_CONV3_C8 = 3 };
enum { E_BF16 = 0, E_CONVT = 1, E_F32_SPLIT = 2, E_TANH_BF16 = 3, E_F32 = 4 };

struct GemmTN {
    const bf16_t* A;
    const bf16_t* B;      // [N][K]
    const float* bias;    // [N] ([Co] for E_CONVT), may be null
    void* out;
    void* out2;           // E_F32_SPLIT: second half
    float* stat_sum;      // [gridM][N] per-workgroup column sums of the stored values, may be null
    float* stat_sq;
    int M, N, K;
    int lda;              // A_PLAIN
    int H, W, C;          // gather geometry (see above)
    int ldo;              // output row stride in elements
    int split;            // E_F32_SPLIT: columns per half
    int Co;               // E_CONVT: channels per tap
    int wshift, hshift;   // log2(W), log2(H) when both are powers of two, else -1
    unsigned a_bytes, b_bytes;   // sizes of A and B in bytes (buffer descriptors; both < 4 GiB)
    int ksteps, nsplit;          // split-K: K-steps per workgroup along z and number of slabs (0, 0 = off)
    double flops, bytes;  // algorithmic work of this launch (profiling only)
};

// GEMM row m -> pixel (img, h, w) of the H x W grid (shifts when the grid is a power of two)
__device__ __forceinline__ void split_px(const GemmTN& p, int m, int& w, int& h, int& img) {
    if (p.W <= 0) { w = h = img = 0; return; }
    if (p.wshift >= 0 && p.hshift >= 0) {
        w = m & (p.W - 1);
        h = (m >> p.wshift) & (p.H - 1);
        img = m >> (p.wshift + p.hshift);
    } else {
        w = m % p.W;
        const int t = m / p.W;
        h = t % p.H;
        img = t / p.H;
    }
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4): four v_add_f32 with DPP operand
// swizzles (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), no LDS traffic.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// s_waitcnt vmcnt(n) for a wave-uniform runtime n in [0, 3 * per]: at most n global->LDS copies stay in flight
template <int PER>
__device__ __forceinline__ void wait_vm_groups(int groups) {
    if (groups <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (groups == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else if (groups == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
}

// NST-deep ring of LDS stages: the copies of K-step ks + NST - 1 are issued while K-step ks computes, so a
// tile has NST - 1 K-steps of compute to arrive (one K-step is shorter than an L2 round trip).
//
// PERSIST (launches of one or two K-steps, no split-K): a workgroup walks the row tiles blockIdx.x, + gridDim.x, ...
// and requests the NEXT tile's operands before it starts the epilogue of the current one.  Such a launch is little
// more than a load phase and a store phase per tile; walked this way the two overlap inside a workgroup instead of
// only between the two or three workgroups that fit a CU.
template <int BM, int BN, int BK, int WAVES_M, int AMODE, int EPI, int NST, bool PERSIST = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const GemmTN p) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int TM = BM / WAVES_M, TN = BN / WAVES_N;
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int ROWB = BK * 2;               // bytes per LDS row
    constexpr int CH = BK / 8;                 // 16-byte chunks per row
    constexpr int RPI = 64 / CH;               // rows written by one wave-instruction
    constexpr int SW = (BK == 64) ? 1 : 2;     // rows per 256-byte bank row = 1 << SW
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int A_INSTR = BM / RPI / 4, B_INSTR = BN / RPI / 4;
    static_assert(FM >= 1 && FN >= 1 && A_INSTR >= 1 && B_INSTR >= 1, "tile too small");
    static_assert(NST >= 2 && NST <= 5, "ring depth");
    static_assert(NST * STAGE >= (int)(2 * WAVES_M * BN * sizeof(float)), "stats scratch does not fit");
    static_assert((NST - 2) * (A_INSTR + B_INSTR) <= 63, "vmcnt is a 6-bit counter");

    __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];
    __shared__ float red_persist[PERSIST ? 2 * WAVES_M * BN : 1];     // statistics scratch that does not alias the stages

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: addresses and branches built on it stay scalar
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    // ---- per-lane staging state ------------------------------------------------------
    // Tiles are fetched with buffer_load ... lds: a 32-bit byte offset per lane against a buffer
    // descriptor held in SGPRs.  An offset past the buffer (OOB below) makes the hardware deliver
    // zeros, which is how padding pixels and rows past M / N are filled -- no pointer select, no
    // 64-bit address arithmetic in the loop.
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
    unsigned a_off[A_INSTR];   // byte offset of this lane's 16-byte chunk at K = 0 (OOB: row >= M)
    int a_hw[A_INSTR];
    int a_chunk[A_INSTR];
    auto setup_a = [&](int mbase) {
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (wave + 4 * i) * RPI + lane / CH;
        const int chunk = (lane % CH) ^ ((row >> SW) & (CH - 1));
        const int m = mbase + row;
        a_hw[i] = 0;
        a_chunk[i] = chunk;
        int w, h, img;
        split_px(p, m, w, h, img);
        unsigned off;
        if (AMODE == A_CONV3_C8) {
            off = (unsigned)m * 16u;
            a_hw[i] = (h << 16) | w;
        } else if (AMODE == A_PLAIN) {
            off = ((unsigned)m * (unsigned)p.lda + chunk * 8) * 2u;
        } else if (AMODE == A_CONV3) {
            off = ((unsigned)m * (unsigned)p.C + chunk * 8) * 2u;
            a_hw[i] = (h << 16) | w;
        } else {
            off = ((((unsigned)img * 2 * p.H + 2 * h) * (2 * p.W) + 2 * w) * (unsigned)p.C + chunk * 8) * 2u;
        }
        a_off[i] = m < p.M ? off : OOB;
        if (AMODE == A_CONV3 || AMODE == A_CONV3_C8) {
            if (m >= p.M) a_hw[i] = 0x7FFF7FFF;     // fails every bounds test below
        }
    }
    };
    setup_a(m0);
    unsigned b_off[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int row = (wave + 4 * i) * RPI + lane / CH;
        const int chunk = (lane % CH) ^ ((row >> SW) & (CH - 1));
        const int n = n0 + row;
        b_off[i] = n < p.N ? ((unsigned)n * (unsigned)p.K + chunk * 8) * 2u : OOB;
    }
    const int cpt = (AMODE == A_PLAIN || AMODE == A_CONV3_C8) ? 1 : p.C / BK;   // K-steps per tap

    auto stage = [&](int buf, int ks) {
        unsigned char* abase = smem + buf * STAGE;
        int tap_off = 0;      // bytes
        int dh = 0, dw = 0;
        if (AMODE == A_PLAIN) {
            tap_off = ks * BK * 2;
        } else if (AMODE == A_CONV3_C8) {
            tap_off = 0;
        } else if (AMODE == A_CONV3) {
            const int tap = ks / cpt, c0 = (ks - tap * cpt) * BK;
            dh = tap / 3 - 1;
            dw = tap % 3 - 1;
            tap_off = ((dh * p.W + dw) * p.C + c0) * 2;
        } else {
            const int tap = ks / cpt, c0 = (ks - tap * cpt) * BK;
            tap_off = (((tap >> 1) * 2 * p.W + (tap & 1)) * p.C + c0) * 2;
        }
#pragma unroll
        for (int i = 0; i < A_INSTR; ++i) {
            unsigned off = a_off[i] == OOB ? OOB : a_off[i] + (unsigned)tap_off;
            if (AMODE == A_CONV3) {
                const int h = (a_hw[i] >> 16) + dh, w = (a_hw[i] & 0xFFFF) + dw;
                if (!((unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)) off = OOB;
            }
            if (AMODE == A_CONV3_C8) {
                const int tp = ks * 8 + a_chunk[i];
                const int eh = tp / 3 - 1, ew = tp % 3 - 1;
                const int h = (a_hw[i] >> 16) + eh, w = (a_hw[i] & 0xFFFF) + ew;
                off = (tp < 9 && (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.W)
                          ? a_off[i] + (unsigned)((eh * p.W + ew) * 16) : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(abase + (wave + 4 * i) * 1024), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const unsigned off = b_off[i] == OOB ? OOB : b_off[i] + (unsigned)(ks * BK * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lptr_t)(abase + A_BYTES + (wave + 4 * i) * 1024), 16, (int)off, 0, 0,
                                                     0);
        }
    };

    f32x4 acc[FM][FN];

    // the bias row is requested before the first tile: a load at the top of the epilogue would expose a memory round
    // trip in every workgroup, and short-K launches are little more than prologue + epilogue
    float bv[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * TN + j * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
        if (p.bias && n < p.N) {
            const float4 b4 = *(const float4*)(p.bias + (EPI == E_CONVT ? n % p.Co : n));
            bv[j][0] = b4.x; bv[j][1] = b4.y; bv[j][2] = b4.z; bv[j][3] = b4.w;
        }
    }
    // split-K (E_F32 only): workgroup z covers K-steps [z * ksteps, ...) and writes slab z of the output
    const int ks0 = p.ksteps > 0 ? (int)blockIdx.z * p.ksteps : 0;
    const int nk = p.ksteps > 0 ? min(p.ksteps, p.K / BK - ks0) : p.K / BK;
#pragma unroll
    for (int q = 0; q < (PERSIST ? NST : NST - 1); ++q)     // PERSIST: nk <= NST, every K-step of a tile has its own stage
        if (q < nk) stage(q, ks0 + q);
    const int ntiles = (p.M + BM - 1) / BM;
    for (int tile = blockIdx.x;; tile += gridDim.x) {
    m0 = tile * BM;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int slot = 0;                     // ring slot of K-step ks
    for (int ks = 0; ks < nk; ++ks) {
        wait_vm_groups<A_INSTR + B_INSTR>(min(NST - 2, nk - 1 - ks));   // everything up to K-step ks has landed
        barrier_keep_copies();            // not __syncthreads(): its fence would drain the younger copies too
        const unsigned char* abase = smem + slot * STAGE;
        const unsigned char* bbase = abase + A_BYTES;
        // all fragments of this K-step are requested up front, the next tile's global->LDS copies are
        // issued while they are in flight, then the matrix instructions drain them in order
        bf16x8 xa[BK / 32][FM], wb[BK / 32][FN];
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            const int chunk = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * TM + i * 16 + (lane & 15);
                xa[kk][i] = *(const bf16x8*)(abase + row * ROWB + ((chunk ^ ((row >> SW) & (CH - 1))) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * TN + j * 16 + (lane & 15);
                wb[kk][j] = *(const bf16x8*)(bbase + row * ROWB + ((chunk ^ ((row >> SW) & (CH - 1))) << 4));
            }
        }
        if (!PERSIST && ks + NST - 1 < nk) stage(slot == 0 ? NST - 1 : slot - 1, ks0 + ks + NST - 1);   // the slot read in K-step ks - 1
        slot = slot + 1 == NST ? 0 : slot + 1;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[kk][j], xa[kk][i], acc[i][j], 0, 0, 0);
    }

    if (PERSIST && tile + (int)gridDim.x < ntiles) {
        barrier_keep_copies();            // every wave has its last fragments: the stages are free
        setup_a(m0 + (int)gridDim.x * BM);
#pragma unroll
        for (int q = 0; q < NST; ++q)
            if (q < nk) stage(q, ks0 + q);
    }

    // ---- epilogue ----------------------------------------------------------------------
    // acc[i][j][r]: pixel m = m0 + wm*TM + i*16 + (lane&15), column n = n0 + wn*TN + j*16 + (lane>>4)*4 + r
    const bool want_stats = p.stat_sum != nullptr;
    float ssum[FN][4], ssq[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[j][r] = ssq[j][r] = 0.f;

#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * TM + i * 16 + (lane & 15);
        const bool mok = m < p.M;
        long orow = 0;
        if (EPI == E_CONVT) {
            int jj, ii, img;
            split_px(p, m, jj, ii, img);
            orow = ((long)img * 2 * p.H + 2 * ii) * (2 * p.W) + 2 * jj;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * TN + j * 16 + (lane >> 4) * 4;
            const bool ok = mok && n < p.N;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r] + bv[j][r];
                if (EPI == E_TANH_BF16) v[r] = tanhf(v[r]);
            }
            if (want_stats && ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { ssum[j][r] += v[r]; ssq[j][r] += v[r] * v[r]; }
            }
            if (!ok) continue;
            if (EPI == E_BF16 || EPI == E_TANH_BF16) {
                *(uint2*)((bf16_t*)p.out + (long)m * p.ldo + n) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            } else if (EPI == E_CONVT) {
                const int tap = n / p.Co, co = n - tap * p.Co;
                const long pix = orow + (long)(tap >> 1) * (2 * p.W) + (tap & 1);
                *(uint2*)((bf16_t*)p.out + pix * p.Co + co) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            } else if (EPI == E_F32_SPLIT) {
                float* dst = n < p.split ? (float*)p.out + (long)m * p.split + n
                                         : (float*)p.out2 + (long)m * p.split + (n - p.split);
                *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                float* o = (float*)p.out + (p.ksteps > 0 ? (long)blockIdx.z * p.M * p.ldo : 0L);
                *(float4*)(o + (long)m * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }

    if (want_stats) {
        // every wave is done reading the staging buffers (PERSIST: own scratch, and no fence that would drain the
        // next tile's copies; the barrier keeps the previous tile's readers of the scratch apart from the writers)
        if (PERSIST) barrier_keep_copies(); else __syncthreads();
        float* red = PERSIST ? red_persist : (float*)smem;   // [2][WAVES_M][BN]
        // all 16-lane sums first (independent DPP chains), then one predicated block of LDS stores
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ssum[j][r] = row16_sum(ssum[j][r]);
                ssq[j][r] = row16_sum(ssq[j][r]);
            }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = wn * TN + j * 16 + (lane >> 4) * 4 + r;
                    red[wm * BN + c] = ssum[j][r];
                    red[(WAVES_M + wm) * BN + c] = ssq[j][r];
                }
        }
        if (PERSIST) barrier_keep_copies(); else __syncthreads();
        if (tid < BN && n0 + tid < p.N) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) { s += red[w * BN + tid]; q += red[(WAVES_M + w) * BN + tid]; }
            p.stat_sum[(long)tile * p.N + n0 + tid] = s;
            p.stat_sq[(long)tile * p.N + n0 + tid] = q;
        }
    }
    if (!PERSIST || tile + (int)gridDim.x >= ntiles) break;
    }   // tile loop
}

template <int BM, int BN, int BK, int WAVES_M, int AMODE, int EPI>
static int launch_one(const GemmTN& p, hipStream_t s) {
    // ring depth: as deep as lets two workgroups share a CU's 160 KB of LDS
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    constexpr int NST = STAGE_BYTES <= 16384 ? 4 : STAGE_BYTES <= 24576 ? 3 : 2;
    // a grid of at most one workgroup per CU has the whole LDS to itself: 5 stages keep 4 K-steps of copies in
    // flight, which is what it takes to cover an HBM round trip with the short K-steps of a small tile
    constexpr int NST_DEEP = STAGE_BYTES * 5 <= 160 * 1024 ? 5 : NST;
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, p.ksteps > 0 ? p.nsplit : 1);
    static char name[64] = "";
    if (!name[0]) snprintf(name, sizeof(name), "gemm_tn_kernel<%d,%d,%d,%d,%d,%d>", BM, BN, BK, WAVES_M, AMODE, EPI);
    ProfScope prof(name, p.flops, p.bytes, s);
    // one or two K-steps: both tiles are requested up front and nothing is refilled, so a deeper ring only costs LDS --
    // two stages leave room for a third (or fifth) workgroup per CU, and these launches live on how many
    // workgroups overlap their load, compute and store phases
    const int nk = p.ksteps > 0 ? p.ksteps : p.K / BK;
    constexpr int PER_CU = 2 * STAGE_BYTES + 4096 <= 160 * 1024 / 3 ? 3 : 2;     // resident workgroups with two stages
    if (p.ksteps <= 0 && nk <= 2 && (long)grid.x * grid.y > 256 * PER_CU) {
        // more tiles than the chip holds at once: resident workgroups walk them (see PERSIST)
        grid.x = (256 * PER_CU + grid.y - 1) / grid.y;
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, BK, WAVES_M, AMODE, EPI, 2, true>), grid, dim3(256), 0, s, p);
    } else if (NST != 2 && nk <= 2)
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, BK, WAVES_M, AMODE, EPI, 2>), grid, dim3(256), 0, s, p);
    else if (NST_DEEP != NST && (long)grid.x * grid.y * grid.z <= 256 && p.K / BK >= 8)
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, BK, WAVES_M, AMODE, EPI, NST_DEEP>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, BK, WAVES_M, AMODE, EPI, NST>), grid, dim3(256), 0, s, p);
    VAE_LAUNCH_CHECK("gemm_tn_kernel");
    return 0;
}

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// rows of the M dimension per workgroup: 128, or 64 when 128-row tiles would leave CUs idle
static int pick_bm(long M, int N) {
    const long bn = N <= 64 ? 64 : 128;
    const long blocks = ((M + 127) / 128) * ((N + bn - 1) / bn);
    return blocks < 256 ? 64 : 128;
}

template <int AMODE, int EPI>
static int launch_bk64(GemmTN p, hipStream_t s) {
    p.wshift = ilog2_exact(p.W);
    p.hshift = ilog2_exact(p.H);
    const bool small = pick_bm(p.M, p.N) == 64;
    if (p.N <= 64) {
        if (small) return launch_one<64, 64, 64, 2, AMODE, EPI>(p, s);
        return launch_one<128, 64, 64, 4, AMODE, EPI>(p, s);
    }
    if (small) {
        // still under one workgroup per CU with 64 x 128 tiles: halve the tile width too
        const long blocks = ((long)(p.M + 63) / 64) * ((p.N + 127) / 128);
        if (blocks < 256) return launch_one<64, 64, 64, 2, AMODE, EPI>(p, s);
        return launch_one<64, 128, 64, 2, AMODE, EPI>(p, s);
    }
    return launch_one<128, 128, 64, 2, AMODE, EPI>(p, s);
}

// a_elems: bf16 elements in the A operand's buffer (the staging offsets are 32-bit byte offsets into it)
static int check_common(GemmTN& p, int bk, const char* who, double a_elems) {
    VAE_CHECK(p.A && p.B && p.out, "%s: null pointer", who);
    const double lim = 4294967296.0 - 64.0;
    VAE_CHECK(2.0 * a_elems < lim && 2.0 * p.N * (double)p.K < lim, "%s: an operand of 4 GiB or more is not supported", who);
    p.a_bytes = (unsigned)(2.0 * a_elems);
    p.b_bytes = (unsigned)(2.0 * p.N * (double)p.K);
    VAE_CHECK(p.M > 0 && p.N > 0 && p.K > 0, "%s: empty problem M=%d N=%d K=%d", who, p.M, p.N, p.K);
    VAE_CHECK(p.K % bk == 0, "%s: K=%d must be a multiple of %d", who, p.K, bk);
    VAE_CHECK(p.N % 4 == 0, "%s: N=%d must be a multiple of 4", who, p.N);
    VAE_CHECK((((uintptr_t)p.A | (uintptr_t)p.B | (uintptr_t)p.out | (uintptr_t)p.out2) & 15) == 0,
              "%s: operands must be 16-byte aligned", who);
    VAE_CHECK((p.stat_sum == nullptr) == (p.stat_sq == nullptr), "%s: stat_sum/stat_sq must both be set or both null", who);
    return 0;
}

}  // namespace vae

using namespace vae;

extern "C" int vae_gemm_rows_per_block(void) { return 64; }   // the smallest tile: an upper bound on partial rows

extern "C" int vae_gemm_stat_rows(long M, int N) {
    const int bm = pick_bm(M, N);
    return (int)((M + bm - 1) / bm);
}

extern "C" int vae_conv3x3_c8_stat_rows(int n_img, int H, int W, int Cout) {
    if (stream_c8_ok(n_img, H, W, Cout)) return stream_c8_rows(n_img, H, W);
    return vae_gemm_stat_rows((long)n_img * H * W, Cout);
}

extern "C" int vae_convt2x2_stat_rows(int n_img, int H, int W, int Cin, int Cout) {
    if (stream_convt_fwd_ok(n_img, H, W, Cin, Cout)) return stream_convt_fwd_rows(n_img, H, W, Cin, Cout);
    return vae_gemm_stat_rows((long)n_img * H * W, 4 * Cout);
}

extern "C" int vae_convt2x2_stat_fold(int n_img, int H, int W, int Cin, int Cout) {
    if (stream_convt_fwd_ok(n_img, H, W, Cin, Cout)) return stream_convt_fwd_fold(Cin, Cout);
    return 4;
}

extern "C" int vae_conv3x3_stat_rows(int n_img, int H, int W, int Cin, int Cout) {
    const long M = (long)n_img * H * W;
    if (conv3_patch_tn_ok(n_img, H, W, Cin, Cout)) return conv3_patch_tn_rows(M, H, W, Cin, Cout);
    return vae_gemm_stat_rows(M, Cout);
}

extern "C" int vae_conv3x3_gemm(const void* x, const void* w_packed, const float* bias, void* y,
                                float* stat_sum, float* stat_sq, int n_img, int H, int W, int Cin,
                                int Cout, void* stream) {
    GemmTN p = {};
    p.A = (const bf16_t*)x; p.B = (const bf16_t*)w_packed; p.bias = bias; p.out = y;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31), "vae_conv3x3_gemm: bad geometry n=%d H=%d W=%d", n_img, H, W);
    VAE_CHECK(H < 65536 && W < 65536, "vae_conv3x3_gemm: H, W must be < 65536");
    VAE_CHECK(Cin % 64 == 0, "vae_conv3x3_gemm: Cin=%d must be a multiple of 64 (use the im2col path below that)", Cin);
    p.M = n_img * H * W; p.N = Cout; p.K = 9 * Cin; p.H = H; p.W = W; p.C = Cin; p.ldo = Cout;
    p.flops = 2.0 * p.M * p.N * p.K;
    p.bytes = 2.0 * ((double)p.M * Cin + (double)p.N * p.K + (double)p.M * Cout);
    if (int e = check_common(p, 64, "vae_conv3x3_gemm", (double)p.M * Cin)) return e;
    // power-of-two images: the patch-staged kernel (conv3_patch_tn.hip)
    if (conv3_patch_tn_ok(n_img, H, W, Cin, Cout))
        return conv3_patch_tn_launch(x, w_packed, bias, y, stat_sum, stat_sq, n_img, H, W, Cin, Cout, p.flops, p.bytes,
                                     (hipStream_t)stream);
    return launch_bk64<A_CONV3, E_BF16>(p, (hipStream_t)stream);
}

extern "C" int vae_conv3x3_c8_gemm(const void* x8, const void* w_packed, const float* bias, void* y,
                                   float* stat_sum, float* stat_sq, int n_img, int H, int W, int Cout,
                                   void* stream) {
    GemmTN p = {};
    p.A = (const bf16_t*)x8; p.B = (const bf16_t*)w_packed; p.bias = bias; p.out = y;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W < (1L << 31) && H < 65536 && W < 65536,
              "vae_conv3x3_c8_gemm: bad geometry n=%d H=%d W=%d", n_img, H, W);
    p.M = n_img * H * W; p.N = Cout; p.K = 128; p.H = H; p.W = W; p.C = 8; p.ldo = Cout;
    p.flops = 2.0 * p.M * p.N * 27.0;      // 3 real input channels x 9 taps
    p.bytes = 2.0 * ((double)p.M * 8 + (double)p.N * p.K + (double)p.M * Cout);
    if (int e = check_common(p, 64, "vae_conv3x3_c8_gemm", (double)p.M * 8)) return e;
    if (stream_c8_ok(n_img, H, W, Cout))
        return stream_c8_launch(x8, w_packed, bias, y, stat_sum, stat_sq, nullptr, nullptr, nullptr, n_img, H, W, Cout, p.flops,
                                p.bytes, (hipStream_t)stream);
    return launch_bk64<A_CONV3_C8, E_BF16>(p, (hipStream_t)stream);
}

// The first conv in two passes: its input is 16 bytes per pixel and its output 128, so convolving twice is cheaper than
// writing the conv output and reading it back for the BatchNorm.
extern "C" int vae_conv3x3_c8_twice_ok(int n_img, int H, int W, int Cout) {
    return n_img > 0 && H > 0 && W > 0 && stream_c8_ok(n_img, H, W, Cout) ? 1 : 0;
}
extern "C" int vae_conv3x3_c8_stats(const void* x8, const void* w_packed, const float* bias, float* stat_sum, float* stat_sq,
                                    int n_img, int H, int W, int Cout, void* stream) {
    VAE_CHECK(x8 && w_packed && stat_sum && stat_sq, "vae_conv3x3_c8_stats: null pointer");
    VAE_CHECK(vae_conv3x3_c8_twice_ok(n_img, H, W, Cout), "vae_conv3x3_c8_stats: geometry n=%d %dx%d Cout=%d not covered (vae_conv3x3_c8_twice_ok)",
              n_img, H, W, Cout);
    VAE_CHECK(((uintptr_t)x8 & 15) == 0 && ((uintptr_t)w_packed & 15) == 0, "vae_conv3x3_c8_stats: operands must be 16-byte aligned");
    const double M = (double)n_img * H * W;
    return stream_c8_launch(x8, w_packed, bias, nullptr, stat_sum, stat_sq, nullptr, nullptr, nullptr, n_img, H, W, Cout,
                            2.0 * M * Cout * 27.0, 2.0 * (M * 8 + (double)Cout * 128), (hipStream_t)stream);
}
extern "C" int vae_conv3x3_c8_bn_act(const void* x8, const void* w_packed, const float* bias, const float* scale,
                                     const float* shift, void* out, void* relu_bits, int n_img, int H, int W, int Cout,
                                     void* stream) {
    VAE_CHECK(x8 && w_packed && scale && shift && out, "vae_conv3x3_c8_bn_act: null pointer");
    VAE_CHECK(vae_conv3x3_c8_twice_ok(n_img, H, W, Cout), "vae_conv3x3_c8_bn_act: geometry n=%d %dx%d Cout=%d not covered (vae_conv3x3_c8_twice_ok)",
              n_img, H, W, Cout);
    VAE_CHECK(((uintptr_t)x8 & 15) == 0 && ((uintptr_t)w_packed & 15) == 0 && ((uintptr_t)out & 15) == 0,
              "vae_conv3x3_c8_bn_act: operands must be 16-byte aligned");
    const double M = (double)n_img * H * W;
    return stream_c8_launch(x8, w_packed, bias, out, nullptr, nullptr, scale, shift, relu_bits, n_img, H, W, Cout,
                            2.0 * M * Cout * 27.0, 2.0 * (M * 8 + (double)Cout * 128 + M * Cout) + (relu_bits ? M * Cout / 8 : 0.0),
                            (hipStream_t)stream);
}

extern "C" int vae_convt2x2_fwd_gemm(const void* x, const void* w_packed, const float* bias, void* y,
                                     float* stat_sum, float* stat_sq, int n_img, int H, int W, int Cin,
                                     int Cout, void* stream) {
    GemmTN p = {};
    p.A = (const bf16_t*)x; p.B = (const bf16_t*)w_packed; p.bias = bias; p.out = y;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W * 4 < (1L << 31), "vae_convt2x2_fwd_gemm: bad geometry");
    VAE_CHECK(Cin % 64 == 0 && Cout % 4 == 0, "vae_convt2x2_fwd_gemm: Cin=%d must be a multiple of 64, Cout=%d of 4", Cin, Cout);
    p.M = n_img * H * W; p.N = 4 * Cout; p.K = Cin; p.lda = Cin; p.H = H; p.W = W; p.Co = Cout;
    p.flops = 2.0 * p.M * p.N * p.K;
    p.bytes = 2.0 * ((double)p.M * Cin + (double)p.N * p.K + (double)p.M * p.N);
    if (int e = check_common(p, 64, "vae_convt2x2_fwd_gemm", (double)p.M * Cin)) return e;
    if (stream_convt_fwd_ok(n_img, H, W, Cin, Cout))
        return stream_convt_fwd_launch(x, w_packed, bias, y, stat_sum, stat_sq, n_img, H, W, Cin, Cout, p.flops, p.bytes,
                                       (hipStream_t)stream);
    return launch_bk64<A_PLAIN, E_CONVT>(p, (hipStream_t)stream);
}

extern "C" int vae_convt2x2_dgrad_gemm(const void* dy, const void* w_packed, void* dx, int n_img, int H,
                                       int W, int Cin, int Cout, void* stream) {
    GemmTN p = {};
    p.A = (const bf16_t*)dy; p.B = (const bf16_t*)w_packed; p.out = dx;
    VAE_CHECK(n_img > 0 && H > 0 && W > 0 && (long)n_img * H * W * 4 < (1L << 31), "vae_convt2x2_dgrad_gemm: bad geometry");
    VAE_CHECK(Cout % 64 == 0, "vae_convt2x2_dgrad_gemm: Cout=%d must be a multiple of 64", Cout);
    p.M = n_img * H * W; p.N = Cin; p.K = 4 * Cout; p.H = H; p.W = W; p.C = Cout; p.ldo = Cin;
    p.flops = 2.0 * p.M * p.N * p.K;
    p.bytes = 2.0 * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N);
    if (int e = check_common(p, 64, "vae_convt2x2_dgrad_gemm", 4.0 * p.M * Cout)) return e;
    return launch_bk64<A_CONVT_DGRAD, E_BF16>(p, (hipStream_t)stream);
}

extern "C" int vae_linear_gemm(const void* a, const void* w, const float* bias, void* out, void* out2,
                               float* stat_sum, float* stat_sq, int M, int N, int K, int epilogue,
                               void* stream) {
    GemmTN p = {};
    p.A = (const bf16_t*)a; p.B = (const bf16_t*)w; p.bias = bias; p.out = out; p.out2 = out2;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldo = N;
    p.flops = 2.0 * M * N * K;
    p.bytes = 2.0 * ((double)M * K + (double)N * K) + (epilogue >= VAE_EPI_F32 ? 4.0 : 2.0) * M * N;
    hipStream_t s = (hipStream_t)stream;
    if (K % 64 != 0) {
        VAE_CHECK(epilogue == VAE_EPI_BF16, "vae_linear_gemm: K=%d not a multiple of 64 is only supported with the bf16 epilogue", K);
        if (int e = check_common(p, 32, "vae_linear_gemm", (double)M * K)) return e;
        if (N <= 64) return launch_one<128, 64, 32, 4, A_PLAIN, E_BF16>(p, s);
        return launch_one<128, 128, 32, 2, A_PLAIN, E_BF16>(p, s);
    }
    if (int e = check_common(p, 64, "vae_linear_gemm", (double)M * K)) return e;
    switch (epilogue) {
        case VAE_EPI_BF16: return launch_bk64<A_PLAIN, E_BF16>(p, s);
        case VAE_EPI_TANH_BF16: return launch_bk64<A_PLAIN, E_TANH_BF16>(p, s);
        case VAE_EPI_F32: return launch_bk64<A_PLAIN, E_F32>(p, s);
        case VAE_EPI_F32_SPLIT:
            VAE_CHECK(out2 && N % 2 == 0 && (N / 2) % 4 == 0, "vae_linear_gemm: split epilogue needs out2 and N/2 a multiple of 4");
            p.split = N / 2;
            return launch_bk64<A_PLAIN, E_F32_SPLIT>(p, s);
        default:
            VAE_CHECK(false, "vae_linear_gemm: unknown epilogue %d", epilogue);
    }
    return 0;
}

// ---- split-K for small-M linear layers --------------------------------------------------------
// With M = batch rows only, a [M][N] output has too few tiles to fill 256 CUs while K is thousands deep
// (fc_mu | fc_logvar forward: 512 x 256 x 4096).  The K range is cut into nsplit slabs computed by
// separate workgroups as fp32 [nsplit][M][N]; the finish kernel adds the slabs and the bias in a fixed
// order (deterministic) and writes the output(s).
namespace vae {
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slab, int nsplit, long mn, int N,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            float* __restrict__ out2, int split) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= mn) return;
    const int n = (int)(e % N);
    const long m = e / N;
    float4 a = bias ? *(const float4*)(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 3 < nsplit; s += 4) {       // four slabs requested at once, added in slab order
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(const float4*)(slab + (long)(s + i) * mn + e);
#pragma unroll
        for (int i = 0; i < 4; ++i) { a.x += v[i].x; a.y += v[i].y; a.z += v[i].z; a.w += v[i].w; }
    }
    for (; s < nsplit; ++s) {
        const float4 v = *(const float4*)(slab + (long)s * mn + e);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (split > 0) {
        float* dst = n < split ? out + m * split + n : out2 + m * split + (n - split);
        *(float4*)dst = a;
    } else {
        *(float4*)(out + e) = a;
    }
}
}  // namespace vae

extern "C" int vae_linear_splitk(int M, int N, int K) {
    if (K % 64 != 0) return 1;
    const long tiles = ((long)(M + 63) / 64) * ((N + 63) / 64);
    const int nk = K / 64;
    long s = (256 + tiles - 1) / tiles;
    if (s > nk / 4) s = nk / 4;
    if (s < 1) s = 1;
    const long per = (nk + s - 1) / s;
    return (int)((nk + per - 1) / per);
}

extern "C" int vae_linear_gemm_splitk(const void* a, const void* w, const float* bias, float* out, float* out2,
                                      float* slab, int nsplit, int M, int N, int K, int epilogue, void* stream) {
    VAE_CHECK(epilogue == VAE_EPI_F32 || epilogue == VAE_EPI_F32_SPLIT, "vae_linear_gemm_splitk: fp32 epilogues only");
    VAE_CHECK(slab && K % 64 == 0, "vae_linear_gemm_splitk: needs a slab and K a multiple of 64");
    VAE_CHECK(nsplit >= 1 && nsplit <= K / 64, "vae_linear_gemm_splitk: nsplit=%d out of range", nsplit);
    VAE_CHECK(N % 4 == 0 && ((uintptr_t)slab & 15) == 0 && ((uintptr_t)out & 15) == 0, "vae_linear_gemm_splitk: alignment");
    int split = 0;
    if (epilogue == VAE_EPI_F32_SPLIT && out) {
        VAE_CHECK(out2 && N % 2 == 0 && (N / 2) % 4 == 0, "vae_linear_gemm_splitk: split epilogue needs out2 and N/2 a multiple of 4");
        split = N / 2;
    }
    GemmTN p = {};
    p.A = (const bf16_t*)a; p.B = (const bf16_t*)w; p.out = slab;
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldo = N;
    const int nk = K / 64;
    p.ksteps = (nk + nsplit - 1) / nsplit;
    p.nsplit = nsplit;
    VAE_CHECK((long)(nsplit - 1) * p.ksteps < nk, "vae_linear_gemm_splitk: nsplit=%d leaves empty splits", nsplit);
    p.flops = 2.0 * M * N * K;
    p.bytes = 2.0 * ((double)M * K + (double)N * K) + 4.0 * M * N * (nsplit + 1.0);
    hipStream_t s = (hipStream_t)stream;
    if (int e = check_common(p, 64, "vae_linear_gemm_splitk", (double)M * K)) return e;
    p.wshift = p.hshift = -1;
    if (int e = launch_one<64, 64, 64, 2, A_PLAIN, E_F32>(p, s)) return e;
    if (!out) return 0;          // the caller's next kernel adds the slabs up (bias first, then in slab order)
    const long mn = (long)M * N;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((mn / 4 + 255) / 256)), dim3(256), 0, s, slab, nsplit, mn, N, bias,
                       out, out2, split);
    VAE_LAUNCH_CHECK("splitk_finish_kernel");
    return 0;
}
