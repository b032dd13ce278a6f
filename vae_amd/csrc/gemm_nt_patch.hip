// 3x3 conv weight gradient, all nine taps per workgroup ("patch" form of the NT GEMM):
//
//   dW[co][tap][ci] = sum_pix dY[pix][co] * X[pix + tap][ci]
//
// A workgroup owns a 64(co) x 64(ci) block of the weight tensor for ALL 9 taps and walks a range of
// 64-pixel K tiles.  Per K tile it stages the dY rows once (8 KB) and the input PATCH once: the
// tile's pixels plus a one-pixel halo, zero where the halo leaves the image (<= 25 KB), instead of
// nine shifted copies of the tile -- 5.6x fewer bytes through L2/LDS per FLOP than one tap per
// workgroup, and dY is read once instead of nine times.  The nine taps are then nine row-shifted
// transposing reads (ds_read_b64_tr_b16) of the same patch.  Each wave owns 16 input channels x
// all 64 output channels x 9 taps (36 MFMA tiles, 144 VGPRs), so the dY fragments are read once
// per 9 taps.  K is split over gridDim.z into slabs [split][co][tap][ci] exactly like the one-tap
// kernel, so the same reducer folds them.
//
// Geometry: a K tile is 64 consecutive pixels = G groups of R image rows (R*W*G = 64); with
// H*W >= 64 that is one group of 64/W rows of one image, for smaller images 64/(H*W) whole images.
// Patch row of pixel (g, r, x), r in [-1, R], x in [-1, W]:  (g*(R+2) + r+1)*(W+2) + x+1.
//
// The bank swizzle of the patch is a function of the patch COLUMN (x+1) only, and the image width
// is a template parameter: a vertical tap shift is then a compile-time byte offset on the same
// address register (the swizzle does not change), a horizontal shift picks one of three
// precomputed addresses.  12 address registers serve the 36 shifted reads of a K tile, which
// leaves room to keep the next tap's fragment in flight while the current one feeds the MFMAs.
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) short4v* l4ptr_t;

struct PatchP {
    const bf16_t* P;   // dY [Ktot][Co]
    const bf16_t* Q;   // X  [Ktot][Ci]
    float* slab;
    int Co, Ci, Ktot;
    int H, wshift;
    int R, G, rwshift;     // rows per group, groups per tile, log2(R*W)
    int PR;                // patch rows
    int ktiles, kps;       // K tiles in total / per split
    long slab_stride;
};

constexpr int PATCH_ROWS_MAX = 200;                   // >= 3*(64+2) = 198, multiple of 8
constexpr int PATCH_INSTR_PER_WAVE = (PATCH_ROWS_MAX / 8 + 3) / 4;   // 7
constexpr int WGRAD_FRAG_AHEAD = 1;                   // input fragments requested ahead of the matrix instructions (<= 3;
                                                      // 2 and 3 measured: 2-5 % on the kernel alone, nothing in the step)
constexpr int ROWB = 128;                             // 64 channels of bf16

// XOR applied to the 16-byte chunk index of a 128-byte LDS row; `i` is the row index (dY tile) or
// the patch column (input patch).  One transposing read touches, per 32-lane half, rows/columns
// {b..b+3} and {b+8..b+11}: bit 0 picks the 128-byte half of the bank line by itself, bits 1 and 3
// spread the rest over the four 32-byte slots.
__device__ __forceinline__ int swz128(int i) { return (((i >> 1) & 1) | (((i >> 3) & 1) << 1)) << 1; }

// Transposing fragment read (two ds_read_b64_tr_b16) through inline asm.  While global->LDS copies are in flight the
// compiler puts s_waitcnt vmcnt(0) in front of every transposing read it issues itself (the copies are pending LDS
// writes to it, and it cannot tell the buffer being filled from the one being read): each round would wait for the
// copies it has just requested for the NEXT round.  The compiler does not count asm loads: frag_wait<N>() is the
// wait (at most N LDS reads still in flight) that must precede the first use of a fragment.
__device__ __forceinline__ bf16x8 tr_pair(unsigned lo_addr, unsigned hi_addr) {
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(lo_addr), "v"(hi_addr) : "memory");
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// the same with an immediate byte offset (a vertical tap shift costs no address arithmetic)
template <int OFF>
__device__ __forceinline__ bf16x8 tr_pair_off(unsigned lo_addr, unsigned hi_addr) {
    short4v lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4"
                 : "=&v"(lo), "=&v"(hi) : "v"(lo_addr), "v"(hi_addr), "n"(OFF) : "memory");
    typedef __attribute__((ext_vector_type(8))) short short8v;
    const short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
// at most N of this wave's LDS reads still in flight; `f` is a fragment about to be used (ties the wait to it)
template <int N>
__device__ __forceinline__ void frag_wait(bf16x8& f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
}
// the same for an n that is a constant only after unrolling
__device__ __forceinline__ void frag_wait_n(int n, bf16x8& f) {
    switch (n) {
        case 0: frag_wait<0>(f); break;
        case 2: frag_wait<2>(f); break;
        case 4: frag_wait<4>(f); break;
        case 6: frag_wait<6>(f); break;
        case 8: frag_wait<8>(f); break;
        case 10: frag_wait<10>(f); break;
        case 12: frag_wait<12>(f); break;
        case 14: frag_wait<14>(f); break;
        case 15: frag_wait<15>(f); break;
        default: frag_wait<0>(f); break;
    }
}

// 512 threads = two groups of four waves.  Both groups hold the same 64 x 9 x 64 accumulator tile and walk the
// workgroup's K range two K tiles at a time (group g takes tiles beg+g, beg+2+g, ...); at the end group 1 hands
// its accumulators to group 0 through LDS, so one slab is written per workgroup: half the split-K bytes of
// two independent 256-thread workgroups at the same number of resident waves.
//
// Measured and not kept: three staging buffers per group with a counted wait (a copy then has two rounds to arrive
// instead of half a round) was 2 % slower -- a round is not waiting for memory; it is the issue of the copies
// (~200 cycles apiece, 9 per wave) plus the fragment reads in front of 72 matrix instructions per wave.
template <int W>
__global__ __launch_bounds__(512, 1) void conv3_wgrad_patch_kernel(const PatchP p) {
    constexpr int W2 = W + 2;
    constexpr int QD = WGRAD_FRAG_AHEAD;
    constexpr int P_BYTES = 64 * ROWB, Q_BYTES = PATCH_ROWS_MAX * ROWB, STAGE = P_BYTES + Q_BYTES;
    constexpr int FOLD_BYTES = 9 * 4 * 4 * 64 * 16;       // one group's accumulators: 36 float4 per lane, 4 waves
    constexpr int SMEM_BYTES = 4 * STAGE > FOLD_BYTES ? 4 * STAGE : FOLD_BYTES;
    constexpr int QI = PATCH_INSTR_PER_WAVE;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);   // wave-uniform
    const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 64, split = blockIdx.z;
    const int kt_beg = split * p.kps + grp, kt_end = min(split * p.kps + p.kps, p.ktiles);   // this group's tiles: stride 2
    const bf16_t* zero = (const bf16_t*)g_zero_page;

    // ---- per-lane patch staging state (the lane's patch rows never change) -------------
    // packed: bits 0-7 pixel offset inside the tile + 128, 8-15 image-row offset r + 1, 16-18 source chunk,
    // bit 24 "row holds a pixel" (else padding: left/right halo or past the patch)
    int q_st[QI];
    const int ninstr = (p.PR + 7) >> 3;
#pragma unroll
    for (int i = 0; i < QI; ++i) {
        const int prow = (wave + 4 * i) * 8 + (lane >> 3);
        const int per = (p.R + 2) * W2;
        const int g = prow / per, rem = prow - g * per;
        const int rr = rem / W2, xx = rem - rr * W2;
        int st = ((lane & 7) ^ swz128(xx)) << 16;
        if (prow < p.PR && xx >= 1 && xx <= W) st |= ((g * p.R + (rr - 1)) * W + (xx - 1) + 128) | (rr << 8) | (1 << 24);
        q_st[i] = st;
    }
    const int p_row0 = wave * 8 + (lane >> 3);
    const int p_ch0 = (lane & 7) ^ swz128(p_row0);

    // copies are buffer_load ... lds: a 32-bit byte offset per lane against a descriptor in SGPRs; an offset past the
    // buffer delivers zeros -- halo pixels, rows outside the image and the K tail cost a select, not a branch and a
    // 64-bit pointer
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const __amdgpu_buffer_rsrc_t p_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.P, 0, (int)((unsigned)p.Ktot * (unsigned)p.Co * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t q_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.Q, 0, (int)((unsigned)p.Ktot * (unsigned)p.Ci * 2u), 0x00020000);
    auto stage = [&](int buf, int kt) {
        unsigned char* pbase = smem + (2 * buf + grp) * STAGE;
        unsigned char* qbase = pbase + P_BYTES;
        const int k0 = kt * 64;
        {
            int ra = p_row0, ca = p_ch0;
            asm volatile("" : "+v"(ra), "+v"(ca));      // keep the address math inside the loop (registers are scarce here)
            const int ka = k0 + ra, kb = ka + 32;       // rows 32 apart agree in both swizzle bits
            const unsigned prow = (unsigned)p.Co * 2;
            const unsigned oa = ka < p.Ktot ? (unsigned)ka * prow + (unsigned)(m0 * 2 + ca * 16) : OOB;
            const unsigned ob = kb < p.Ktot ? (unsigned)kb * prow + (unsigned)(m0 * 2 + ca * 16) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, (lptr_t)(pbase + wave * 1024), 16, (int)oa, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, (lptr_t)(pbase + (wave + 4) * 1024), 16, (int)ob, 0, 0, 0);
        }
        const unsigned rowb = (unsigned)p.Ci * 2;
        const int h0 = p.G == 1 ? ((k0 >> p.wshift) & (p.H - 1)) : 0;
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            if (wave + 4 * i < ninstr) {     // wave-uniform
                int st = q_st[i];
                asm volatile("" : "+v"(st));
                const int k = k0 + (st & 255) - 128, r = ((st >> 8) & 255) - 1;
                unsigned off = OOB;
                if ((st >> 24) && (unsigned)(h0 + r) < (unsigned)p.H && (unsigned)k < (unsigned)p.Ktot)
                    off = (unsigned)k * rowb + (unsigned)(n0 * 2 + ((st >> 12) & 0x70));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (lptr_t)(qbase + (wave + 4 * i) * 1024), 16, (int)off, 0, 0, 0);
            }
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- per-lane fragment addresses (byte offsets inside a stage) ----------------------
    // lanes 16g..16g+15 own pixels kk*32 + 8g .. +7 of the tile: quad `lo` = the first four, `hi` = the
    // last four; lane t of the group reads pixel (t >> 2) of the quad, channels 4(t & 3).. of the fragment.
    const int t16 = lane & 15;
    int q_addr[2][2][3];       // [kk][lo/hi][dw + 1], tap row dh = -1; (dh + 1) * W2 * ROWB is added as an immediate
    int p_addr[2][2];          // [kk][lo/hi] for output-channel fragment j = 0; fragment j flips chunk bits 1-2
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            const int k = kk * 32 + 8 * (lane >> 4) + 4 * hl;             // first pixel of the quad
            const int g = k >> p.rwshift, kr = k & ((1 << p.rwshift) - 1);
            const int r = kr >> p.wshift, x = (kr & (W - 1)) + (t16 >> 2);
            const int prow = (g * (p.R + 2) + r) * W2 + x + 1;            // patch row of the pixel one image row up
            const int chunk = wave * 2 + ((t16 & 3) >> 1), within = (t16 & 1) * 8;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int xx = x + d;                                    // patch column of the pixel shifted by dw = d - 1
                q_addr[kk][hl][d] = P_BYTES + (prow + d - 1) * ROWB + (((chunk ^ swz128(xx)) << 4) | within);
            }
            const int row = k + (t16 >> 2);
            p_addr[kk][hl] = row * ROWB + (((((t16 & 3) >> 1)) ^ swz128(row)) << 4 | within);
        }

    const unsigned smem_u = (unsigned)(size_t)(lptr_t)smem;      // LDS byte address of the staging area
    auto q_frag = [&](unsigned sbase, int kk, int t) -> bf16x8 {
        const int dh = t / 3, dw = t % 3;      // 0..2
        return tr_pair(sbase + q_addr[kk][0][dw] + dh * W2 * ROWB, sbase + q_addr[kk][1][dw] + dh * W2 * ROWB);
    };

    if (kt_beg < kt_end) stage(0, kt_beg);
    const int niter = (min(split * p.kps + p.kps, p.ktiles) - split * p.kps + 1) >> 1;   // same for both groups
    for (int it = 0; it < niter; ++it) {
        const int kt = kt_beg + 2 * it;
        const int buf = it & 1, nbuf = buf ^ 1;
        // not __syncthreads(): nothing but the copies of this round is in flight here, and the fragment reads below are
        // asm, so no fence is needed beyond the two waits
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        barrier_keep_copies();
        // The copies for the next round cost the issuing wave ~200 cycles apiece (9 per wave).  Each SIMD holds one wave
        // of either group: group 0 issues its copies now, group 1 after its first 36 MFMAs, so that one group's matrix
        // instructions run under the other's copy issue instead of all eight waves issuing, then all computing.
        if (grp == 0 && kt + 2 < kt_end) stage(nbuf, kt + 2);
        const unsigned sbase = smem_u + (unsigned)((2 * buf + grp) * STAGE);
        const bool live = kt < kt_end;                    // odd tile count: group 1 only copies in the last round
        // dY fragments of both halves of the tile: the second half's are requested while the first half's last taps
        // run, so only the first read of a round exposes its LDS latency.  LDS reads complete in order: a fragment is
        // ready once no more reads are in flight than were issued after it.
        bf16x8 pb[2][4];
        auto load_pb = [&](int kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int alo = p_addr[kk][0], ahi = p_addr[kk][1];
                if (j) {   // the XOR stays in the loop: four address registers instead of sixteen
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(alo) : "n"(j << 5), "v"(p_addr[kk][0]));
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ahi) : "n"(j << 5), "v"(p_addr[kk][1]));
                }
                pb[kk][j] = tr_pair(sbase + alo, sbase + ahi);
            }
        };
        // The 18 (half tile, tap) steps of a round are 4 MFMAs = 64 cycles each, less than an LDS round trip under load:
        // the input fragment of step s + QD is requested before step s runs.  LDS reads complete in order, so the
        // fragment of step s is ready once no more reads are in flight than were issued after it: two per later
        // fragment, and the eight of load_pb(1), which goes out QD steps before tap 6 of the first half so that it is
        // older than every fragment of the second half.
        bf16x8 qa[QD + 1];
        if (live) {
            load_pb(0);
#pragma unroll
            for (int d = 0; d < QD; ++d) qa[d] = q_frag(sbase, d / 9, d % 9);
        }
#pragma unroll
        for (int st = 0; st < 18; ++st) {
            if (live) {
                const int kk = st / 9, t = st % 9;
                if (st + QD < 18) qa[(st + QD) % (QD + 1)] = q_frag(sbase, (st + QD) / 9, (st + QD) % 9);
                if (st == 6 - QD) load_pb(1);
                const int later = 17 - st < QD ? 17 - st : QD;
                frag_wait_n(2 * later + (st >= 6 - QD && st <= 6 ? 8 : 0), qa[st % (QD + 1)]);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[st % (QD + 1)], pb[kk][j], acc[t][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            if (st == 8 && grp == 1 && kt + 2 < kt_end) stage(nbuf, kt + 2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // The two groups hold partial sums of the same tile.  They swap halves through LDS (the staging buffers are
    // dead): group 0 ends up with the complete taps 0-4, group 1 with taps 5-8, and each stores its share of the
    // slab -- eight waves fold and store instead of four.  (a + b == b + a bit for bit: the order is immaterial.)
    __syncthreads();
    float4* fold = (float4*)smem + (wave * 64 + lane);
    if (grp == 0) {
#pragma unroll
        for (int t = 5; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fold[((t - 5) * 4 + j) * 256] = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
    } else {
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                fold[(16 + t * 4 + j) * 256] = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
    }
    __syncthreads();
    // acc[t][j][r]: ci = n0 + wave*16 + (lane>>4)*4 + r ; co = m0 + j*16 + (lane&15)
    float* out = p.slab + (long)split * p.slab_stride;
    const int ci = n0 + wave * 16 + (lane >> 4) * 4;
    if (grp == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = m0 + j * 16 + (lane & 15);
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const float4 o = fold[(16 + t * 4 + j) * 256];
                *(float4*)(out + ((long)co * 9 + t) * p.Ci + ci) =
                    make_float4(acc[t][j][0] + o.x, acc[t][j][1] + o.y, acc[t][j][2] + o.z, acc[t][j][3] + o.w);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = m0 + j * 16 + (lane & 15);
#pragma unroll
            for (int t = 5; t < 9; ++t) {
                const float4 o = fold[((t - 5) * 4 + j) * 256];
                *(float4*)(out + ((long)co * 9 + t) * p.Ci + ci) =
                    make_float4(acc[t][j][0] + o.x, acc[t][j][1] + o.y, acc[t][j][2] + o.z, acc[t][j][3] + o.w);
            }
        }
    }
}


// Producer / consumer form of the same kernel.  In the form above every wave issues its share of the copies AND
// runs its share of the matrix instructions, one after the other: a CU takes in only ~70 GB/s through global->LDS
// copies, a 1 KB copy holds the issuing wave for ~100-200 cycles, and with two waves per SIMD doing the same work the
// copy issue (0.9 us per round), the fragment reads and the MFMAs (0.55 us) simply add up.  Here waves 4-7 only copy
// and waves 0-3 only compute, two 64-pixel K tiles per barrier out of two pairs of stages: ONE accumulator set per
// workgroup (no hand-over between groups), and the copies of round q + 1 run under the matrix instructions of round q.
constexpr int PC_STAGE = 64 * ROWB + PATCH_ROWS_MAX * ROWB, PC_SMEM_BYTES = 4 * PC_STAGE;
static_assert(PC_SMEM_BYTES <= 160 * 1024 && 4 * 64 * 16 * 16 <= PC_SMEM_BYTES, "LDS");

// the workgroup's job: weight block (ci from n0, co from m0) of problem p, K tiles [split * kps, + kps)
// ABL (diagnostic builds only, -DVAE_ABLATE): bit 0 no matrix instructions, bit 1 no copies, bit 2 no fragment reads
template <int W, int ABL = 0>
__device__ __forceinline__ void wgrad_pc_body(const PatchP& p, int n0, int m0, int split, unsigned char* smem) {
    constexpr int W2 = W + 2;
    constexpr int QD = 2;     // input fragments requested ahead (4 measured: 10 % slower)
    constexpr int P_BYTES = 64 * ROWB, STAGE = PC_STAGE;
    constexpr int QI = PATCH_INSTR_PER_WAVE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int role = __builtin_amdgcn_readfirstlane(tid >> 8), wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);   // 0: compute, 1: copy
    const int kt_beg = split * p.kps, kt_end = min(kt_beg + p.kps, p.ktiles);
    const int ntile = kt_end - kt_beg;

    // A copy holds its wave -- and the issue port of its SIMD, which the computing wave there needs for its matrix
    // instructions -- for as long as its address arithmetic takes.  Everything that does not depend on the K tile is
    // worked out here, once per lane and copy slot, so that a copy costs two adds, two compares and a select (no
    // multiply, no branch).  q_off: byte offset of the lane's 16 bytes for K tile 0; q_pix: pixel offset inside the tile
    // (a padding row gets one that is out of range for every tile); q_r: image-row offset.
    int q_off[QI], q_pix[QI], q_r[QI];
    const int ninstr = (p.PR + 7) >> 3;
    const unsigned rowb = (unsigned)p.Ci * 2;
#pragma unroll
    for (int i = 0; i < QI; ++i) {
        const int prow = (wave + 4 * i) * 8 + (lane >> 3);
        const int per = (p.R + 2) * W2;
        const int g = prow / per, rem = prow - g * per;
        const int rr = rem / W2, xx = rem - rr * W2;
        const bool pixel = prow < p.PR && xx >= 1 && xx <= W;
        const int pix = (g * p.R + (rr - 1)) * W + (xx - 1);
        q_pix[i] = pixel ? pix : (int)0xC0000000;          // k0 + this is never in [0, Ktot)
        q_r[i] = rr - 1;
        q_off[i] = pix * (int)rowb + n0 * 2 + (((lane & 7) ^ swz128(xx)) << 4);
    }
    const int p_row0 = wave * 8 + (lane >> 3);
    const unsigned prow_b = (unsigned)p.Co * 2;
    const int p_off0 = p_row0 * (int)prow_b + m0 * 2 + (((lane & 7) ^ swz128(p_row0)) << 4);
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const __amdgpu_buffer_rsrc_t p_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.P, 0, (int)((unsigned)p.Ktot * (unsigned)p.Co * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t q_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.Q, 0, (int)((unsigned)p.Ktot * (unsigned)p.Ci * 2u), 0x00020000);
    // a tile past the range is all zero fill
    auto stage = [&](int buf, int kt) {
        if (ABL & 2) return;
        unsigned char* pbase = smem + buf * STAGE;
        unsigned char* qbase = pbase + P_BYTES;
        const int k0 = kt * 64;
        const unsigned ktot = kt < kt_end ? (unsigned)p.Ktot : 0u;      // wave-uniform
        {
            const unsigned base = (unsigned)k0 * prow_b;                // scalar
            const int ka = k0 + p_row0;                                 // rows 32 apart agree in both swizzle bits
            const unsigned oa = (unsigned)ka < ktot ? base + (unsigned)p_off0 : OOB;
            const unsigned ob = (unsigned)(ka + 32) < ktot ? base + 32u * prow_b + (unsigned)p_off0 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, (lptr_t)(pbase + wave * 1024), 16, (int)oa, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, (lptr_t)(pbase + (wave + 4) * 1024), 16, (int)ob, 0, 0, 0);
        }
        const unsigned base = (unsigned)k0 * rowb;                      // scalar
        const int h0 = p.G == 1 ? ((k0 >> p.wshift) & (p.H - 1)) : 0;
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            if (wave + 4 * i < ninstr) {     // wave-uniform
                const bool ok = ((unsigned)(h0 + q_r[i]) < (unsigned)p.H) & ((unsigned)(k0 + q_pix[i]) < ktot);
                const unsigned off = ok ? base + (unsigned)q_off[i] : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(q_rsrc, (lptr_t)(qbase + (wave + 4 * i) * 1024), 16, (int)off, 0, 0, 0);
            }
        }
    };

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int t16 = lane & 15;
    int q_addr[2][2][3];       // [kk][lo/hi][dw + 1], tap row dh = -1
    int p_addr[2][2];          // [kk][lo/hi] for output-channel fragment j = 0
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
            const int k = kk * 32 + 8 * (lane >> 4) + 4 * hl;             // first pixel of the quad
            const int g = k >> p.rwshift, kr = k & ((1 << p.rwshift) - 1);
            const int r = kr >> p.wshift, x = (kr & (W - 1)) + (t16 >> 2);
            const int prow = (g * (p.R + 2) + r) * W2 + x + 1;            // patch row of the pixel one image row up
            const int chunk = wave * 2 + ((t16 & 3) >> 1), within = (t16 & 1) * 8;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int xx = x + d;
                q_addr[kk][hl][d] = P_BYTES + (prow + d - 1) * ROWB + (((chunk ^ swz128(xx)) << 4) | within);
            }
            const int row = k + (t16 >> 2);
            p_addr[kk][hl] = row * ROWB + (((((t16 & 3) >> 1)) ^ swz128(row)) << 4 | within);
        }
    const unsigned smem_u = (unsigned)(size_t)(lptr_t)smem;
    // The two roles run separate loops with the same number of barriers.  A round is two K tiles (stages 2 (q & 1) and
    // 2 (q & 1) + 1): after barrier q the copying waves refill the pair that round q - 1 used with the tiles of round
    // q + 1, which then have a whole round of matrix instructions to land.
    const int nround = (ntile + 1) >> 1;
    if (role == 1) {
        stage(0, kt_beg);
        stage(1, kt_beg + 1);
        for (int q = 0; q < nround; ++q) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // round q has landed
            barrier_keep_copies();                                // ... for every producer; the consumers are done with round q - 1
            const int nb = 2 * ((q + 1) & 1);
            stage(nb, kt_beg + 2 * q + 2);
            stage(nb + 1, kt_beg + 2 * q + 3);
        }
    } else {
        for (int q = 0; q < nround; ++q) {
            barrier_keep_copies();
            const bool two = 2 * q + 1 < ntile;                   // odd tile count: the last round has one tile
            const unsigned sbase = smem_u + (unsigned)(2 * (q & 1) * STAGE);
            // A single computing wave per SIMD is bound by instruction issue: the stage base is folded into the twelve
            // fragment addresses once per round; the vertical tap shift and the second tile of the pair (+ STAGE) are
            // immediate offsets instead of adds in front of every read.
            unsigned qb[2][2][3], pbs[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl) {
                    pbs[kk][hl] = sbase + p_addr[kk][hl];
#pragma unroll
                    for (int d = 0; d < 3; ++d) qb[kk][hl][d] = sbase + q_addr[kk][hl][d];
                }
            // step s of a round = (tile s / 18, half tile (s / 9) & 1, tap s % 9)
            auto q_frag = [&](int s) -> bf16x8 {
                const int tl = s / 18, kk = (s / 9) & 1, t = s % 9, dh = t / 3, dw = t % 3;
                const unsigned lo = qb[kk][0][dw], hi = qb[kk][1][dw];
                if (ABL & 4) return __builtin_bit_cast(bf16x8, make_uint4(lo, hi, lo, hi));
                if (tl == 0) {
                    if (dh == 0) return tr_pair_off<0>(lo, hi);
                    if (dh == 1) return tr_pair_off<W2 * ROWB>(lo, hi);
                    return tr_pair_off<2 * W2 * ROWB>(lo, hi);
                }
                if (dh == 0) return tr_pair_off<STAGE>(lo, hi);
                if (dh == 1) return tr_pair_off<STAGE + W2 * ROWB>(lo, hi);
                return tr_pair_off<STAGE + 2 * W2 * ROWB>(lo, hi);
            };
            bf16x8 pb[2][4];
            auto load_pb = [&](int half, int j0) {   // half = 2 * tile + kk, fragments j0, j0 + 1 into pb[half & 1]
                const int kk = half & 1;
#pragma unroll
                for (int j = j0; j < j0 + 2; ++j) {
                    unsigned alo = pbs[kk][0], ahi = pbs[kk][1];
                    if (j) {
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(alo) : "n"(j << 5), "v"(pbs[kk][0]));
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(ahi) : "n"(j << 5), "v"(pbs[kk][1]));
                    }
                    if (ABL & 4) pb[kk][j] = __builtin_bit_cast(bf16x8, make_uint4(alo, ahi, alo, ahi));
                    else pb[kk][j] = half < 2 ? tr_pair_off<0>(alo, ahi) : tr_pair_off<STAGE>(alo, ahi);
                }
            };
            bf16x8 qa[QD + 1];
            load_pb(0, 0);
            load_pb(0, 2);
#pragma unroll
            for (int d = 0; d < QD; ++d) qa[d] = q_frag(d);
            // The dY fragments of the next half tile go out in two batches of four reads, at taps TA and TB of the current
            // half, each right after that step's input fragment.  LDS reads complete in order: the fragment of step s
            // is ready once no more reads are in flight than were issued after it -- two per later input fragment and
            // four per batch issued in steps s - QD .. s (a count above the counter's 15 is clamped: a stricter wait).
            // A batch issued at tap T is complete at the wait of tap T + QD + 1 <= 8, before the next half starts.
            constexpr int TA = 1, TB = 3;
            static_assert(TB + QD + 1 <= 8, "dY fragments must land before their half tile starts");
#pragma unroll
            for (int st = 0; st < 36; ++st) {
                const int half = st / 9, t = st % 9;
                // (requests that run into a second tile that does not exist read stale LDS and are never used)
                if (st + QD < 36) qa[(st + QD) % (QD + 1)] = q_frag(st + QD);
                if (half < 3 && t == TA) load_pb(half + 1, 0);
                if (half < 3 && t == TB) load_pb(half + 1, 2);
                const int later = 35 - st < QD ? 35 - st : QD;
                int newer = 2 * later;
                if (half < 3 && t >= TA && t <= TA + QD) newer += 4;
                if (half < 3 && t >= TB && t <= TB + QD) newer += 4;
                frag_wait_n(newer > 15 ? 15 : newer, qa[st % (QD + 1)]);
                if (ABL & 1) {      // keep the fragments alive without the matrix pipe
                    if (st == 35) acc[0][0][0] += __builtin_bit_cast(f32x4, __builtin_bit_cast(uint4, qa[st % (QD + 1)]) ^ __builtin_bit_cast(uint4, pb[half & 1][0]))[0];
                } else if (st < 18 || two) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[st % (QD + 1)], pb[half & 1][j], acc[t][j], 0, 0, 0);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the computing waves hand taps 5-8 to the copying waves through LDS (the stages are dead): eight waves store
    float4* xch = (float4*)smem + (wave * 64 + lane);
    if (role == 0) {
#pragma unroll
        for (int t = 5; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xch[((t - 5) * 4 + j) * 256] = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
    }
    __syncthreads();
    // acc[t][j][r]: ci = n0 + wave*16 + (lane>>4)*4 + r ; co = m0 + j*16 + (lane&15)
    float* out = p.slab + (long)split * p.slab_stride;
    const int ci = n0 + wave * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = m0 + j * 16 + (lane & 15);
        if (role == 0) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
                *(float4*)(out + ((long)co * 9 + t) * p.Ci + ci) = make_float4(acc[t][j][0], acc[t][j][1], acc[t][j][2], acc[t][j][3]);
        } else {
#pragma unroll
            for (int t = 5; t < 9; ++t) *(float4*)(out + ((long)co * 9 + t) * p.Ci + ci) = xch[((t - 5) * 4 + j) * 256];
        }
    }
}

template <int W>
__global__ __launch_bounds__(512, 1) void conv3_wgrad_pc_kernel(const PatchP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[PC_SMEM_BYTES];
    wgrad_pc_body<W>(p, blockIdx.x * 64, blockIdx.y * 64, blockIdx.z, smem);
}
#ifdef VAE_ABLATE
template <int ABL>
__global__ __launch_bounds__(512, 1) void conv3_wgrad_pc_abl_kernel(const PatchP p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[PC_SMEM_BYTES];
    wgrad_pc_body<32, ABL>(p, blockIdx.x * 64, blockIdx.y * 64, blockIdx.z, smem);
}
#endif
int g_wgrad_ablate = 0;       // diagnostic builds: which parts of the 32-wide producer / consumer kernel are left out

// Several weight gradients in ONE launch.  Every 512-thread workgroup of these kernels keeps a 64 x 9 x 64 fp32 accumulator
// tile (147 KB) and leaves it as a split-K slab, so a launch that fills the 256 CUs writes 37.7 MB of slabs whatever the
// layer is -- launched one layer at a time, the seven conv layers of the encoder wrote 264 MB of slabs per backward pass
// for 16 MB of gradients, and the fold read them all back.  Here the workgroups of one launch are dealt out over the
// layers in proportion to their work (a job = one layer; the planner gives every workgroup about the same number of K
// tiles), so the LAUNCH writes 37.7 MB, each workgroup walks a K range several times as long (prologue, accumulator
// hand-over and slab store are paid once per workgroup, not once per layer) and a backward stage has one launch
// boundary instead of one per layer.  The price: every layer's dY must stay alive until the stage's launch.
struct WgradJob {
    PatchP p;
    int W, wg0, nx, ny;       // image width (selects the instantiation), first workgroup, ci / co tiles
};
struct WgradJobs {
    WgradJob j[VAE_MAX_WGRAD_JOBS];
    int n;
};
__global__ __launch_bounds__(512, 1) void conv3_wgrad_multi_kernel(const WgradJobs jobs) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[PC_SMEM_BYTES];
    int ji = 0;
    for (int i = 1; i < jobs.n; ++i)
        if ((int)blockIdx.x >= jobs.j[i].wg0) ji = i;       // uniform over the workgroup
    const WgradJob& jb = jobs.j[ji];
    const int local = (int)blockIdx.x - jb.wg0;
    // tiles fastest.  (Measured and not kept: blocks of 8 K ranges x all tiles with the K range fastest, so that the
    // tiles of a range -- which read the same dY and X rows -- share one XCD's L2: the step +0.5 % at batch 512, equal at
    // the 64-wide configuration; the re-reads are served by the memory-side cache either way.)
    const int nxi = local % jb.nx, t = local / jb.nx, nyi = t % jb.ny, split = t / jb.ny;
    switch (jb.W) {
        case 4: wgrad_pc_body<4>(jb.p, nxi * 64, nyi * 64, split, smem); break;
        case 8: wgrad_pc_body<8>(jb.p, nxi * 64, nyi * 64, split, smem); break;
        case 16: wgrad_pc_body<16>(jb.p, nxi * 64, nyi * 64, split, smem); break;
        case 32: wgrad_pc_body<32>(jb.p, nxi * 64, nyi * 64, split, smem); break;
        default: wgrad_pc_body<64>(jb.p, nxi * 64, nyi * 64, split, smem); break;
    }
}

// Which form runs (tuning key "wgrad_pc"): 0 by the K tiles per workgroup, 1 always the producer / consumer form, 2 never.
// Measured at batch 512, 32x32 (32 K tiles per workgroup): the kernels alone 10-12 % faster (47 -> 41.5, 45 -> 39.5 us),
// 0.258 -> 0.237 ms per step between events, but the step itself 7 us SLOWER (1.582 -> 1.589 ms, repeatedly, same box);
// at batch 1024, 64x64 (128 K tiles per workgroup) 1.348 -> 1.23 ms and the step 8.49 -> 8.41 ms.  Hence the threshold.
int g_wgrad_pc_mode = 0;
static bool wgrad_use_pc(int kps) { return g_wgrad_pc_mode == 1 || (g_wgrad_pc_mode == 0 && kps >= 64); }

static int ilog2x(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

}  // namespace vae

using namespace vae;

// 1 when the patch kernel covers this geometry (power-of-two image with 4 <= W <= 64, whole K tiles
// inside images or whole images inside K tiles, channel counts multiples of 64)
extern "C" int vae_conv3_wgrad_patch_ok(int H, int W, int Cin, int Cout) {
    if (ilog2x(H) < 0 || ilog2x(W) < 0 || W < 4 || W > 64) return 0;
    if (Cin % 64 || Cout % 64) return 0;
    const long hw = (long)H * W;
    if (hw >= 64 ? 64 % W != 0 : 64 % hw != 0) return 0;
    const int G = hw >= 64 ? 1 : (int)(64 / hw), R = hw >= 64 ? 64 / W : H;
    return G * (R + 2) * (W + 2) <= PATCH_ROWS_MAX ? 1 : 0;      // the patch of a K tile must fit its LDS buffer
}

extern "C" int vae_conv3_wgrad_patch_ksplit(int n_img, int H, int W, int Cin, int Cout) {
    const long ktiles = ((long)n_img * H * W + 63) / 64;
    const long tiles = (long)(Cin / 64) * (Cout / 64);
    long s = (256 + tiles - 1) / tiles;          // one 512-thread workgroup per CU
    const long maxs = ktiles / 8 > 0 ? ktiles / 8 : 1;
    if (s > maxs) s = maxs;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    const long kps = (ktiles + s - 1) / s;
    return (int)((ktiles + kps - 1) / kps);
}

// geometry part of the kernel argument (everything but the K split)
static int make_patch_p(PatchP* p, const void* dy, const void* x, float* slab, int n_img, int H, int W, int Cin, int Cout,
                        const char* who) {
    VAE_CHECK(dy && x && slab, "%s: null pointer", who);
    VAE_CHECK(n_img > 0 && vae_conv3_wgrad_patch_ok(H, W, Cin, Cout),
              "%s: geometry n=%d %dx%d Cin=%d Cout=%d not covered (use vae_wgrad_gemm)", who, n_img, H, W, Cin, Cout);
    VAE_CHECK((long)n_img * H * W * (Cin > Cout ? Cin : Cout) * 2 < (1L << 32), "%s: operand larger than 4 GiB", who);
    VAE_CHECK((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)slab) & 15) == 0, "%s: operands must be 16-byte aligned", who);
    *p = PatchP{};
    p->P = (const bf16_t*)dy; p->Q = (const bf16_t*)x; p->slab = slab;
    p->Co = Cout; p->Ci = Cin; p->Ktot = n_img * H * W;
    p->H = H; p->wshift = ilog2x(W);
    const long hw = (long)H * W;
    if (hw >= 64) { p->G = 1; p->R = 64 / W; } else { p->G = (int)(64 / hw); p->R = H; }
    p->rwshift = ilog2x(p->R * W);
    p->PR = p->G * (p->R + 2) * (W + 2);
    VAE_CHECK(p->PR <= PATCH_ROWS_MAX && p->rwshift >= 0, "%s: internal geometry error", who);
    p->ktiles = (p->Ktot + 63) / 64;
    p->slab_stride = 9L * Cout * Cin;
    return 0;
}
static int set_split(PatchP* p, int nsplit, const char* who) {
    VAE_CHECK(nsplit >= 1 && nsplit <= p->ktiles, "%s: nsplit=%d out of range (1..%d)", who, nsplit, p->ktiles);
    p->kps = (p->ktiles + nsplit - 1) / nsplit;
    VAE_CHECK((long)(nsplit - 1) * p->kps < p->ktiles, "%s: nsplit=%d leaves empty splits", who, nsplit);
    return 0;
}

extern "C" int vae_conv3_wgrad_patch(const void* dy, const void* x, float* slab, int n_img, int H, int W, int Cin,
                                     int Cout, int nsplit, void* stream) {
    PatchP p;
    if (int e = make_patch_p(&p, dy, x, slab, n_img, H, W, Cin, Cout, "vae_conv3_wgrad_patch")) return e;
    if (int e = set_split(&p, nsplit, "vae_conv3_wgrad_patch")) return e;
    dim3 grid(Cin / 64, Cout / 64, nsplit);
    hipStream_t s = (hipStream_t)stream;
    const double flops = 2.0 * Cout * Cin * 9.0 * p.Ktot;
    const double bytes = 2.0 * p.Ktot * ((double)Cout + Cin) + 4.0 * nsplit * 9.0 * Cout * Cin;
    ProfScope prof("conv3_wgrad_patch_kernel", flops, bytes, s);
#ifdef VAE_ABLATE
    if (g_wgrad_ablate && W == 32) {
        switch (g_wgrad_ablate) {
            case 1: hipLaunchKernelGGL(conv3_wgrad_pc_abl_kernel<1>, grid, dim3(512), 0, s, p); break;
            case 2: hipLaunchKernelGGL(conv3_wgrad_pc_abl_kernel<2>, grid, dim3(512), 0, s, p); break;
            case 3: hipLaunchKernelGGL(conv3_wgrad_pc_abl_kernel<3>, grid, dim3(512), 0, s, p); break;
            case 4: hipLaunchKernelGGL(conv3_wgrad_pc_abl_kernel<4>, grid, dim3(512), 0, s, p); break;
            case 5: hipLaunchKernelGGL(conv3_wgrad_pc_abl_kernel<5>, grid, dim3(512), 0, s, p); break;
            case 6: hipLaunchKernelGGL(conv3_wgrad_pc_abl_kernel<6>, grid, dim3(512), 0, s, p); break;
            default: hipLaunchKernelGGL(conv3_wgrad_pc_abl_kernel<7>, grid, dim3(512), 0, s, p); break;
        }
        VAE_LAUNCH_CHECK("conv3_wgrad_pc_abl_kernel");
        return 0;
    }
#endif
#define GO(WW)                                                                                          \
    if (wgrad_use_pc(p.kps)) hipLaunchKernelGGL(conv3_wgrad_pc_kernel<WW>, grid, dim3(512), 0, s, p);      \
    else hipLaunchKernelGGL(conv3_wgrad_patch_kernel<WW>, grid, dim3(512), 0, s, p)
    switch (W) {
        case 4: GO(4); break;
        case 8: GO(8); break;
        case 16: GO(16); break;
        case 32: GO(32); break;
        default: GO(64); break;
    }
#undef GO
    VAE_LAUNCH_CHECK("conv3_wgrad_patch_kernel");
    return 0;
}

// K splits of the jobs of one batched launch: every workgroup gets about the same number of K tiles and the launch has
// at most `max_wgs` workgroups (one per CU: a second round of workgroups would double its time).
extern "C" int vae_conv3_wgrad_multi_plan(const vae_wgrad_job* jobs, int njobs, int max_wgs, int* nsplit_out) {
    VAE_CHECK(jobs && nsplit_out && njobs > 0 && njobs <= VAE_MAX_WGRAD_JOBS, "vae_conv3_wgrad_multi_plan: njobs=%d out of range (1..%d)",
              njobs, VAE_MAX_WGRAD_JOBS);
    if (max_wgs <= 0) max_wgs = 256;
    long tiles[VAE_MAX_WGRAD_JOBS], ktiles[VAE_MAX_WGRAD_JOBS], total = 0, min_wgs = 0;
    for (int i = 0; i < njobs; ++i) {
        const vae_wgrad_job& q = jobs[i];
        VAE_CHECK(q.n_img > 0 && vae_conv3_wgrad_patch_ok(q.H, q.W, q.Cin, q.Cout),
                  "vae_conv3_wgrad_multi_plan: job %d: geometry n=%d %dx%d Cin=%d Cout=%d not covered", i, q.n_img, q.H, q.W, q.Cin, q.Cout);
        tiles[i] = (long)(q.Cin / 64) * (q.Cout / 64);
        ktiles[i] = ((long)q.n_img * q.H * q.W + 63) / 64;
        total += tiles[i] * ktiles[i];
        min_wgs += tiles[i];
    }
    if (max_wgs < min_wgs) max_wgs = (int)min_wgs;       // more output tiles than CUs: one split each
    long kps = (total + max_wgs - 1) / max_wgs;
    if (kps < 8) kps = 8;                                // a split of fewer than 8 K tiles is all prologue and slab store
    for (;; ++kps) {
        long wgs = 0;
        for (int i = 0; i < njobs; ++i) {
            const long k = kps < ktiles[i] ? kps : ktiles[i];
            const long per = (ktiles[i] + k - 1) / k;                       // splits asked for
            const long kk = (ktiles[i] + per - 1) / per;                    // K tiles per split, evened out
            nsplit_out[i] = (int)((ktiles[i] + kk - 1) / kk);
            wgs += tiles[i] * nsplit_out[i];
        }
        if (wgs <= max_wgs) return 0;
    }
}

extern "C" int vae_conv3_wgrad_patch_multi(const vae_wgrad_job* jobs, int njobs, void* stream) {
    VAE_CHECK(jobs && njobs > 0 && njobs <= VAE_MAX_WGRAD_JOBS, "vae_conv3_wgrad_patch_multi: njobs=%d out of range (1..%d)", njobs,
              VAE_MAX_WGRAD_JOBS);
    WgradJobs wj;
    wj.n = njobs;
    int wgs = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < njobs; ++i) {
        const vae_wgrad_job& q = jobs[i];
        WgradJob& d = wj.j[i];
        if (int e = make_patch_p(&d.p, q.dy, q.x, q.slab, q.n_img, q.H, q.W, q.Cin, q.Cout, "vae_conv3_wgrad_patch_multi")) return e;
        if (int e = set_split(&d.p, q.nsplit, "vae_conv3_wgrad_patch_multi")) return e;
        d.W = q.W; d.wg0 = wgs; d.nx = q.Cin / 64; d.ny = q.Cout / 64;
        wgs += d.nx * d.ny * q.nsplit;
        flops += 2.0 * q.Cout * q.Cin * 9.0 * d.p.Ktot;
        bytes += 2.0 * d.p.Ktot * ((double)q.Cout + q.Cin) + 4.0 * q.nsplit * 9.0 * q.Cout * q.Cin;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("conv3_wgrad_patch_kernel", flops, bytes, s);
    hipLaunchKernelGGL(conv3_wgrad_multi_kernel, dim3(wgs), dim3(512), 0, s, wj);
    VAE_LAUNCH_CHECK("conv3_wgrad_multi_kernel");
    return 0;
}
