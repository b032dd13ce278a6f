// 3x3 stride-1 pad-1 convolution (forward, and input-gradient with flipped/transposed weights) as a
// bf16 MFMA implicit GEMM whose activation operand is staged ONCE per 64-channel chunk:
//
//   out[pix][n] = bias[n] + sum_{tap, c} X[pix + tap][c] * Wp[n][tap][c]
//
// A workgroup owns 128 consecutive NHWC pixels x BN output channels.  For every 64-channel chunk
// it copies the pixels' PATCH -- the tile plus a one-pixel halo, zeros where the halo leaves the
// image -- into LDS once, and the nine taps read it at nine row offsets.  The one-tap-per-K-step
// kernel re-fetches a shifted copy of the tile for every tap (9 x 16 KB per chunk); here the
// global->LDS traffic per chunk is the patch (<= 37 KB) plus nine weight tiles, which moves the
// kernel off the L2->LDS fill rate (~70 GB/s per CU) that bounds the one-tap form.
//
// Patch geometry: a tile is G groups of R image rows (R*W*G = 128): H*W >= 128 -> one group of 128/W
// rows of one image, smaller images -> 128/(H*W) whole images.  Patch row of pixel (g, r, x), r in
// [-1, R], x in [-1, W]:  (g*(R+2) + r+1)*(W+2) + x+1, 128 bytes (64 channels) per row.  The bank
// swizzle is a function of the patch column only and W is a template parameter, so a vertical tap
// shift is an immediate offset on the fragment address and a horizontal one selects one of three
// precomputed addresses.
//
// MFMA orientation as in the one-tap kernel: weights are the "A" operand (rows = output channels),
// pixels the "B" operand (columns), so a lane holds 4 consecutive channels of a pixel per
// accumulator: 8-byte bf16 NHWC stores, per-channel batch statistics reduce over 16 lanes.
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct PatchTN {
    const bf16_t* A;      // activations NHWC [M][C]
    const bf16_t* B;      // packed weights [N][9][C]
    const float* bias;    // [N] or null
    bf16_t* out;          // [M][N]
    float* stat_sum;      // [M/128][N] or null
    float* stat_sq;
    int M, N, C;
    int H, wshift;
    int R, G, rwshift;    // rows per group, groups per tile, log2(R*W)
    int PR;               // patch rows
    float inv_per;        // 1 / ((R + 2) * (W + 2))
    unsigned long long* dbg;   // diagnostic builds only: per-workgroup cycle stamps (16 per workgroup), else null
    int ws64_copy_first;       // conv3_ws64 (A/B aid): every wave requests the next patch at the top of the tile
};

// pixels per tile: 128, or 256 with the 64-channel tile (the mirror image of 128 pixels x 128 channels: the same
// 64 x 64 wave tiles, and the per-tile costs -- weight copies, barriers, prologue, statistics -- paid half as often)
constexpr int PT_ROWB = 128;
// patch rows reserved in LDS per image width (the largest patch any covered H gives, rounded up to 8):
// W=4: 8 images of 4x4 = 8*6*6; W=8: 2 images of 8x8 = 2*10*10; W=16: 2 images of 4x16 = 2*6*18;
// W=32: 4 rows = 6*34; W=64: 2 rows = 4*66
// 256-pixel tiles: 4 images of 8x8 = 4*10*10; one 16x16 image = 18*18 (rounded up to 8); 8 rows of 32 = 10*34; 4 rows of 64 = 6*66
constexpr int pt_rows(int W, int BM = 128) {
    return BM == 128 ? (W == 4 ? 288 : W == 8 ? 200 : W == 16 ? 216 : W == 32 ? 208 : 264)
                     : (W == 8 ? 400 : W == 16 ? 328 : W == 32 ? 344 : 400);
}

// s_waitcnt vmcnt(groups * PER) for a wave-uniform runtime group count in [0, 2]
template <int PER>
__device__ __forceinline__ void pt_wait_groups(int groups) {
    if (groups <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (groups == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
}

__device__ __forceinline__ float pt_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// XOR applied to the 16-byte chunk index of a patch row; a function of the patch COLUMN only, so that a vertical tap
// shift is an immediate offset.  ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous -- lanes
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- i.e. a group holds pixels 0-3 and 12-15 of a fragment
// with one chunk and pixels 4-11 with the next.  Bit 0 of the chunk therefore already separates the two pixel sets,
// the row parity (rows are 128 B, the bank row 256 B) separates odd from even pixels, and bits 1-2 must be distinct on
// columns {c, c+2, c+12, c+14} and on {c+4, .., c+10}: (col >> 1) & 3 is.  (The earlier (col >> 1) & 7 also flipped bit
// 0 and collided with the group structure: 40-50 % of the LDS cycles of these reads were bank conflicts by the
// hardware counter.)  4-wide images put four image rows in a fragment and keep 2-way conflicts on half the reads.
__device__ __forceinline__ int pt_swz(int col) { return ((col >> 1) & 3) << 1; }

// workgroups that share a CU: three of the 64-channel form (a third of the LDS and at most 168 VGPRs each), else two
constexpr int pt_wgs(int W, int BN, int BM = 128) { return BN == 64 && BM == 128 && W != 4 ? 3 : 2; }

// KS = 2 ("split K inside the workgroup"): 512 threads = two groups of four waves that own the SAME output tile; group g
// runs the loop below over the channel chunks c = g, g + 2, ... with its own patch buffer and weight ring, group 1 hands
// its accumulators to group 0 through LDS and group 0 runs the epilogue.  For launches with so few tiles that a CU gets
// one four-wave workgroup (one wave per SIMD: every LDS read -> MFMA latency chain is exposed) this puts two waves
// on each SIMD and halves the chain (the 4x4 layers: 36 K-steps -> 18).  The sum over k is then split in two partial
// sums, so results differ from the KS = 1 form in the last bit.
template <int W, int BN, int PT_BM, int KS = 1>
__global__ __launch_bounds__(256 * KS, KS == 2 ? 2 : pt_wgs(W, BN, PT_BM)) void conv3_patch_tn_kernel(const PatchTN p) {
    static_assert(PT_BM == 128 || (PT_BM == 256 && BN == 64 && W >= 8), "tile heights: 128, or 256 with the 64-channel tile");
    static_assert(KS == 1 || PT_BM == 128, "the split form uses 128-pixel tiles");
    constexpr int W2 = W + 2;
    constexpr int WAVES_M = BN == 128 ? 2 : 4, WAVES_N = 4 / WAVES_M;
    constexpr int TM = PT_BM / WAVES_M, TN = BN / WAVES_N;      // wave tile: TM pixels x TN channels
    constexpr int FM = TM / 16, FN = TN / 16;
    constexpr int Q_BYTES = pt_rows(W, PT_BM) * PT_ROWB, B_BYTES = BN * PT_ROWB;
    constexpr int B_INSTR = BN / 8 / 4;
    constexpr int Q_INSTR = (pt_rows(W, PT_BM) / 8 + 3) / 4;    // patch copies per wave
    // weight tiles sit in a ring as deep as the workgroup's share of the CU's 160 KB allows, at most 4
    constexpr int LDS_SHARE = (163840 / (KS == 2 ? 2 : pt_wgs(W, BN, PT_BM))) / 1280 * 1280;     // whole allocation granules
    constexpr int AUX_BYTES = 2048 + BN * 4;                       // statistics accumulators + the bias row
    constexpr int DEPTH = (LDS_SHARE - AUX_BYTES - Q_BYTES) / B_BYTES >= 4 ? 4 : (LDS_SHARE - AUX_BYTES - Q_BYTES) / B_BYTES;
    static_assert(TN == 64, "wave tile is 64 channels wide");
    static_assert(DEPTH >= 2, "ring depth");

    constexpr int RED_BYTES = 2 * WAVES_M * BN * 4;                // statistics accumulators
    static_assert(RED_BYTES + BN * 4 == AUX_BYTES, "auxiliary LDS size");
    constexpr int GROUP_BYTES = Q_BYTES + DEPTH * B_BYTES;        // one group's patch + weight ring
    static_assert(KS == 1 || 256 * 16 * 16 <= KS * GROUP_BYTES, "the accumulator hand-over does not fit");
    __shared__ __attribute__((aligned(16))) unsigned char smem_all[KS * GROUP_BYTES + AUX_BYTES];

    const int lane = threadIdx.x & 63;
    const int grp = KS == 2 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : 0;
    const int tid = threadIdx.x & 255;                            // index inside the group
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: addresses and branches built on it stay scalar
    unsigned char* const smem = smem_all + grp * GROUP_BYTES;     // this group's staging buffers
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int n0 = blockIdx.y * BN;
    int m0 = 0, h0 = 0;          // first pixel of the current tile and its image row (set per tile below)
    const int K = 9 * p.C;

    // ---- patch staging: each lane's patch rows are fixed for the lifetime of the workgroup ----
    // packed: bits 0-9 pixel offset inside the tile + 256, 10-17 image-row offset r + 1, 18-20 source chunk,
    // bit 24 "row holds a pixel" (else padding: left/right halo or past the patch)
    int q_st[Q_INSTR];
    const int ninstr = (p.PR + 7) >> 3;
#pragma unroll
    for (int i = 0; i < Q_INSTR; ++i) {
        const int prow = (wave + 4 * i) * 8 + (lane >> 3);
        // patch row -> (group, image row, column); exact float reciprocal division (operands < 2^9)
        const int per = (p.R + 2) * W2;
        const int g = p.G == 1 ? 0 : (int)(((float)prow + 0.5f) * p.inv_per);
        const int rem = prow - g * per;
        const int rr = (int)(((float)rem + 0.5f) * (1.0f / W2)), xx = rem - rr * W2;
        int st = ((lane & 7) ^ pt_swz(xx)) << 18;
        if (prow < p.PR && xx >= 1 && xx <= W) st |= ((g * p.R + (rr - 1)) * W + (xx - 1) + 256) | (rr << 10) | (1 << 24);
        q_st[i] = st;
    }
    const unsigned rowb = (unsigned)p.C * 2;
    // copies are buffer_load ... lds: 32-bit byte offsets against descriptors in SGPRs; an offset past the
    // buffer (OOB) delivers zeros -- the halo, rows past M and weight rows past N cost no pointer select
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((unsigned)p.M * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((unsigned)p.N * (unsigned)K * 2u), 0x00020000);

    auto stage_patch = [&](int c0) {
#pragma unroll
        for (int i = 0; i < Q_INSTR; ++i) {
            if (wave + 4 * i < ninstr) {     // wave-uniform
                int st = q_st[i];
                asm volatile("" : "+v"(st));      // unpack inside the loop: one register per instruction
                const int m = m0 + (st & 1023) - 256, r = ((st >> 10) & 255) - 1;
                unsigned off = OOB;
                if ((st >> 24) && (unsigned)(h0 + r) < (unsigned)p.H && (unsigned)m < (unsigned)p.M)
                    off = (unsigned)m * rowb + (unsigned)(c0 * 2 + ((st >> 14) & 0x70));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(smem + (wave + 4 * i) * 1024), 16, (int)off, 0, 0, 0);
            }
        }
    };

    // ---- weight tile staging -------------------------------------------------------------------
    unsigned b_off[B_INSTR];     // byte offset of this lane's 16-byte chunk for tap 0, channel chunk 0 (OOB: row >= N)
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int row = (wave + 4 * i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        b_off[i] = n0 + row < p.N ? ((unsigned)(n0 + row) * (unsigned)K + chunk * 8) * 2u : OOB;
    }
    auto stage_b = [&](int slot, int koff) {      // koff = tap * C + c0
        unsigned char* bbase = smem + Q_BYTES + slot * B_BYTES;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const unsigned off = b_off[i] == OOB ? OOB : b_off[i] + (unsigned)(koff * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lptr_t)(bbase + (wave + 4 * i) * 1024), 16, (int)off, 0, 0, 0);
        }
    };

    // ---- fragment addresses --------------------------------------------------------------------
    // [pixel fragment][dw + 1] for the first 32 channels of the chunk (kk = 0), tap row dh = -1; (dh + 1) * W2 * 128 is
    // an immediate; the second 32 channels flip bit 6 (chunk index bit 2), one XOR per read instead of a register
    int a_addr[FM][3];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * TM + i * 16 + (lane & 15);
        const int g = ml >> p.rwshift, kr = ml & ((1 << p.rwshift) - 1);
        const int r = kr >> p.wshift, x = kr & (W - 1);
        const int prow = (g * (p.R + 2) + r) * W2 + x + 1;            // the pixel one image row up
#pragma unroll
        for (int d = 0; d < 3; ++d)
            a_addr[i][d] = (prow + d - 1) * PT_ROWB + (((lane >> 4) ^ pt_swz(x + d)) << 4);
    }
    int b_addr[FN];
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const int row = wn * TN + j * 16 + (lane & 15);
        b_addr[j] = Q_BYTES + row * PT_ROWB + (((lane >> 4) ^ ((row >> 1) & 7)) << 4);
    }

    // ---- persistent loop over pixel tiles: the address tables above are computed once per workgroup ----
    const int nchunk_all = p.C / 64;
    const int nchunk = KS == 2 ? nchunk_all / 2 : nchunk_all;     // this group's chunks (the launcher checks that they split evenly)
    const int ntiles = (p.M + PT_BM - 1) / PT_BM;
    const int nsteps = 9 * nchunk;                 // K-steps per tile and group: one (channel chunk, tap) each
    int slot = 0;                                  // ring slot of the current K-step's weight tile
    // per-channel statistics of the stored values accumulate over all the tiles of this workgroup in a small LDS
    // array [sum | sum of squares][wave row][channel] (each slot has one owning lane: no atomics, no barrier) and
    // leave as ONE partial row per workgroup
    const bool want_stats = p.stat_sum != nullptr;
    float* red = (float*)(smem_all + KS * GROUP_BYTES);
    if (grp == 0)
        for (int i = tid; i < 2 * WAVES_M * BN; i += 256) red[i] = 0.f;     // first read is many barriers away
    // the bias row sits in LDS for the life of the workgroup: a global load at the top of every epilogue would expose
    // a memory round trip per tile
    float* bias_lds = red + 2 * WAVES_M * BN;
    if (grp == 0 && tid < BN) bias_lds[tid] = (p.bias && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (tile != (int)blockIdx.x) barrier_keep_copies();  // the previous tile's patch and weight tiles are done with
    m0 = tile * PT_BM;
    h0 = p.G == 1 ? ((m0 >> p.wshift) & (p.H - 1)) : 0;
    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int cbase = grp * 64;                    // this group's first channel chunk; its chunks are KS * 64 apart
    stage_patch(cbase);
#pragma unroll
    for (int q = 0; q < DEPTH - 1; ++q)            // K-steps 0 .. DEPTH-2 (nsteps >= 9 > DEPTH)
        stage_b((slot + q) % DEPTH, q * p.C + cbase);
    for (int c = 0; c < nchunk; ++c) {
        const int c0 = cbase + c * KS * 64;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // K-step s = 9c + tap needs its weight tile (and, on the first tap of a chunk, the patch, which was
            // issued last): everything but the copies of the next min(DEPTH-2, steps left) K-steps must have landed
            const int left = nsteps - 1 - (9 * c + tap);
            if (tap == 0 && c > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else pt_wait_groups<B_INSTR>(left < DEPTH - 2 ? left : DEPTH - 2);
            barrier_keep_copies();
            {   // refill the slot read in the previous K-step with the tile of K-step s + DEPTH - 1
                const int t2 = tap + DEPTH - 1;
                const int tapn = t2 >= 9 ? t2 - 9 : t2, cn = t2 >= 9 ? c + 1 : c;
                if (cn < nchunk) stage_b((slot + DEPTH - 1) % DEPTH, tapn * p.C + cbase + cn * KS * 64);
            }
            const int dh = tap / 3, dw = tap % 3;
            const int boff = slot * B_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 xa[FM], wb[FN];
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    int a = a_addr[i][dw];
                    if (kk) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(a) : "v"(a_addr[i][dw]));   // stays in the loop
                    xa[i] = *(const bf16x8*)(smem + a + dh * W2 * PT_ROWB);
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    int b = b_addr[j];
                    if (kk) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(b) : "v"(b_addr[j]));
                    wb[j] = *(const bf16x8*)(smem + b + boff);
                }
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
            }
            slot = slot + 1 == DEPTH ? 0 : slot + 1;
        }
        if (c + 1 < nchunk) {
            barrier_keep_copies();            // every wave has read the last tap of this chunk's patch
            stage_patch(c0 + KS * 64);
        }
    }

    // ---- epilogue: bias, bf16 store ---------------------------------------------------------------
    // acc[i][j][r]: pixel m = m0 + wm*TM + i*16 + (lane&15), channel n = n0 + wn*TN + j*16 + (lane>>4)*4 + r
    float bv[FN][4];     // (re)loaded per tile: 16 registers that need not live through the K loop
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        const float4 b4 = *(const float4*)(bias_lds + wn * TN + j * 16 + (lane >> 4) * 4);
        bv[j][0] = b4.x; bv[j][1] = b4.y; bv[j][2] = b4.z; bv[j][3] = b4.w;
    }
    float ssum[FN][4], ssq[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ssum[j][r] = ssq[j][r] = 0.f;
    // The accumulator layout gives a lane 4 channels of one pixel: stored directly, a wave-instruction would touch
    // sixteen 32-byte pieces of sixteen rows.  The tile goes through LDS instead ([pixel][BN] bf16, rows padded by
    // 16 bytes so the 16 pixel-lanes of a write land on different banks) and leaves as 16 bytes per lane in
    // whole rows.
    constexpr int OROW = BN * 2 + 16;
    static_assert(PT_BM * OROW <= Q_BYTES + DEPTH * B_BYTES, "output tile does not fit the staging buffers");
    barrier_keep_copies();                     // every wave is done reading the patch and the weight ring
    if (KS == 2) {
        // group 1 -> group 0: one float4 per (fragment, lane), laid out so that a wave's store is 1 KB contiguous
        float4* hand = (float4*)smem_all + tid;
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    hand[(i * FN + j) * 256] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        barrier_keep_copies();
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const float4 o = hand[(i * FN + j) * 256];
                    acc[i][j][0] += o.x; acc[i][j][1] += o.y; acc[i][j][2] += o.z; acc[i][j][3] += o.w;
                }
        }
        barrier_keep_copies();                 // the hand-over area is the output staging area below
    }
    const bool full = m0 + PT_BM <= p.M && n0 + BN <= p.N;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * TM + i * 16 + (lane & 15);
        const int m = m0 + ml;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int nl = wn * TN + j * 16 + (lane >> 4) * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[j][r];
            if (full || (m < p.M && n0 + nl < p.N)) {      // `full` is uniform: no per-element predicate on whole tiles
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ssum[j][r] += v[r];
                    ssq[j][r] += v[r] * v[r];
                }
            }
            if (grp == 0) *(uint2*)(smem + ml * OROW + nl * 2) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        }
    }
    barrier_keep_copies();
    {
        constexpr int CPR = BN / 8;      // 16-byte chunks per row
#pragma unroll
        for (int q = 0; q < PT_BM * CPR / 256; ++q) {
            const int cidx = q * 256 + tid;
            const int row = cidx / CPR, ch = cidx % CPR;
            const int m = m0 + row, n = n0 + ch * 8;
            if (grp == 0 && m < p.M && n < p.N) *(uint4*)(p.out + (long)m * p.N + n) = *(const uint4*)(smem + row * OROW + ch * 16);
        }
    }
    if (want_stats && grp == 0) {
        // all 16-lane sums first (32 independent DPP chains the scheduler can interleave), then one predicated block
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ssum[j][r] = pt_row16_sum(ssum[j][r]);
                ssq[j][r] = pt_row16_sum(ssq[j][r]);
            }
        if ((lane & 15) == 0) {                      // the one lane that owns a (wave row, channel) slot
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cc = wn * TN + j * 16 + (lane >> 4) * 4 + r;
                    // ds_add_f32 without return: fire and forget (a read-modify-write would wait ~100 cycles per slot)
                    (void)__hip_atomic_fetch_add(&red[wm * BN + cc], ssum[j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void)__hip_atomic_fetch_add(&red[(WAVES_M + wm) * BN + cc], ssq[j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
        }
    }
    }   // tile loop

    if (want_stats) {
        barrier_keep_copies();
        if (grp == 0 && tid < BN && n0 + tid < p.N) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_M; ++w) { s += red[w * BN + tid]; q += red[(WAVES_M + w) * BN + tid]; }
            p.stat_sum[(long)blockIdx.x * p.N + n0 + tid] = s;
            p.stat_sq[(long)blockIdx.x * p.N + n0 + tid] = q;
        }
    }
}


// ============================================================================================================
// The 8-wave form: 256 pixels x 128 output channels per tile, ONE workgroup of 512 threads per CU.
//
// Why a second form.  In the 4-wave kernel above every wave issues ~5 global->LDS copies per K-step (4 weight pieces
// and its share of the patch) beside 32 MFMAs; a copy costs 100-185 cycles of issue, so a wave needs ~1400 cycles per
// K-step for 512 cycles of matrix work and two waves per SIMD cannot close that gap.  Here the weight tile of a K-step
// (16 KB) serves twice the pixels and is copied by eight waves: 2 weight pieces + at most 1 patch piece per wave per
// K-step.  The two waves that share a SIMD (wave w and w + 4) run the K-step in opposite program order -- waves 0-3
// issue their copies first and then compute, waves 4-7 compute first and copy afterwards -- so one's copy issue sits
// under the other's MFMAs with a single barrier per K-step.
//
// The K-steps of all the tiles a workgroup walks form ONE stream: step s reads ring slot s % 3 (= tap % 3: nine taps
// per chunk) and, once its barrier is passed, requests the weight tile of step s + 2 -- which may belong to the next
// tile -- and one eighth-of-a-row-block piece of the NEXT patch (next 64-channel chunk, or the next tile's first
// chunk) into the other of two patch buffers.  Every wave issues the same number of copies per tap (a piece that does
// not exist is a zero-fill into a dummy kilobyte), so the counted waits are compile-time constants: before the
// barrier of a step, everything but the copies issued during the previous step must have landed.
// The epilogue stages the bf16 tile through the patch buffer that was just consumed, 128 pixels at a time, while the
// next tile's first patch and weight tiles are already in flight.
// Per-channel statistics accumulate in registers over all the tiles of the workgroup and are folded once at its end.
constexpr int WD_BM = 256, WD_BN = 128, WD_QP = 7;

// LDS accesses of the epilogue go through inline asm: while global->LDS copies are in flight the compiler puts
// s_waitcnt vmcnt(0) in front of every LDS access it can see (the copies are pending LDS writes to it), which would
// drain the next tile's prefetch and, worse, wait for the global stores of the first half-tile to complete.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void lds_store_b64(unsigned addr, unsigned lo, unsigned hi) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// four 16-byte reads and the wait for them in ONE statement (the compiler does not count asm loads)
__device__ __forceinline__ void lds_load4_b128(unsigned a0, unsigned a1, unsigned a2, unsigned a3, u32x4& r0, u32x4& r1,
                                               u32x4& r2, u32x4& r3) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                 : "memory");
}

// STAMPS: diagnostic instantiation that records the cycle counter at the seams of a workgroup's first tiles
template <int W, bool STAMPS>
__global__ __launch_bounds__(512, 2) void conv3_wide_kernel(const PatchTN p) {
    constexpr int W2 = W + 2;
    constexpr int QROWS = pt_rows(W, WD_BM);
    constexpr int Q_BYTES = QROWS * PT_ROWB, B_BYTES = WD_BN * PT_ROWB;
    constexpr int QPIECES = QROWS / 8;
    static_assert(QROWS % 8 == 0 && QPIECES <= 8 * WD_QP, "patch pieces per wave");
    constexpr int RING = 2 * Q_BYTES, RED = RING + 3 * B_BYTES, DUMMY = RED + 2 * 4 * WD_BN * 4, TOTAL = DUMMY + 1024;
    static_assert(TOTAL <= 163840, "LDS budget");
    constexpr int OROW = WD_BN * 2 + 16;
    static_assert(128 * OROW <= Q_BYTES, "output half-tile does not fit a patch buffer");
    // v_mfma_f32_16x16x32_bf16: 4 x 4 output fragments of 16 x 16 per wave tile, two k-steps of 32 per 64-channel chunk
    // (the 32x32x16 form was measured 6-14 % slower here and spills)
    constexpr int F = 16, NF = 4, NE = 4, NKS = 2;
    typedef f32x4 acc_t;
    __shared__ __attribute__((aligned(16))) unsigned char smem[TOTAL];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // diagnostic build: wave 0 of every workgroup stamps the cycle counter at the seams of its first tiles
    int nstamp = 0;
    auto stamp = [&]() {
        if (STAMPS && p.dbg && wave == 0 && nstamp < 16) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) p.dbg[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + nstamp] = t;
            ++nstamp;
        }
    };
    stamp();                                   // 0: kernel entry
    const int grp = wave >> 2;                 // 0: copies first, 1: copies last
    const int wm = wave >> 1, wn = wave & 1;   // wave tile: pixels [64 wm, +64) x channels [64 wn, +64)
    const int n0 = blockIdx.y * WD_BN;
    const int K = 9 * p.C;
    const int nchunk = p.C / 64;
    const int ntiles = p.M / WD_BM;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const unsigned rowb = (unsigned)p.C * 2;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((unsigned)p.M * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((unsigned)p.N * (unsigned)K * 2u), 0x00020000);
    auto tile_h0 = [&](int m0t) { return p.G == 1 ? ((m0t >> p.wshift) & (p.H - 1)) : 0; };

    // ---- weight pieces: rows [8 (wave + 8 i), +8) of the 128-row tile, i = 0, 1
    unsigned b_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (wave + 8 * i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        b_off[i] = ((unsigned)(n0 + row) * (unsigned)K + chunk * 8) * 2u;
    }
    auto stage_b = [&](int slot, int koff) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lptr_t)(smem + RING + slot * B_BYTES + (wave + 8 * i) * 1024), 16,
                                                     (int)(b_off[i] + (unsigned)(koff * 2)), 0, 0, 0);
    };
    // the weight tiles of steps 0 and 1 go out before anything else is computed: they need no other set-up
    int tile = blockIdx.x;
    stage_b(0, 0);
    stage_b(1, p.C);       // tap 1, chunk 0

    // ---- patch pieces of this wave: piece (wave + 8 i) = patch rows [8 (wave + 8 i), +8), lane -> (row, 16-byte chunk)
    int q_st[WD_QP];
#pragma unroll
    for (int i = 0; i < WD_QP; ++i) {
        const int prow = (wave + 8 * i) * 8 + (lane >> 3);
        const int per = (p.R + 2) * W2;
        const int g = p.G == 1 ? 0 : (int)(((float)prow + 0.5f) * p.inv_per);
        const int rem = prow - g * per;
        const int rr = (int)(((float)rem + 0.5f) * (1.0f / W2)), xx = rem - rr * W2;
        int st = ((lane & 7) ^ pt_swz(xx)) << 18;
        if (prow < p.PR && xx >= 1 && xx <= W) st |= ((g * p.R + (rr - 1)) * W + (xx - 1) + 256) | (rr << 10) | (1 << 24);
        q_st[i] = st;
    }
    // piece i of the patch of (tile first pixel m0t, its image row h0t, channel offset c0) into patch buffer `buf`;
    // valid = 0 (nothing follows) or a piece past the patch: a zero-fill into the dummy kilobyte, so that every
    // wave issues the same number of copies
    auto stage_piece = [&](int i, int buf, int m0t, int h0t, int c0, bool valid) {
        int st = q_st[i];
        asm volatile("" : "+v"(st));
        const int m = m0t + (st & 1023) - 256, r = ((st >> 10) & 255) - 1;
        const bool ok = valid && (st >> 24) && (unsigned)(h0t + r) < (unsigned)p.H && (unsigned)m < (unsigned)p.M;
        const unsigned off = ok ? (unsigned)m * rowb + (unsigned)(c0 * 2 + ((st >> 14) & 0x70)) : OOB;
        const int piece = wave + 8 * i;
        const int dst = piece < QPIECES ? buf * Q_BYTES + piece * 1024 : DUMMY;      // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(smem + dst), 16, (int)off, 0, 0, 0);
    };
    {
        const int m0t = tile * WD_BM, h0t = tile_h0(m0t);
#pragma unroll
        for (int i = 0; i < WD_QP; ++i) stage_piece(i, 0, m0t, h0t, 0, true);
    }

    // ---- fragment addresses (patch buffer 0; += / -= Q_BYTES when the buffers swap).  Fragment of F rows: lane ->
    // row (lane % F), 16-byte chunk (lane / F) of k-step 0; k-step 1 flips address bit 6 (its chunks are 4 further)
    int a_addr[NF][3];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int ml = wm * 64 + i * F + (lane & (F - 1));
        const int g = ml >> p.rwshift, kr = ml & ((1 << p.rwshift) - 1);
        const int r = kr >> p.wshift, x = kr & (W - 1);
        const int prow = (g * (p.R + 2) + r) * W2 + x + 1;            // the pixel one image row up
#pragma unroll
        for (int d = 0; d < 3; ++d)
            a_addr[i][d] = (prow + d - 1) * PT_ROWB + (((lane / F) ^ pt_swz(x + d)) << 4);
    }
    int b_addr[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int row = wn * 64 + j * F + (lane & (F - 1));
        b_addr[j] = RING + row * PT_ROWB + (((lane / F) ^ ((row >> 1) & 7)) << 4);
    }
    // accumulator register e of channel fragment j: channel chan(j, e) of the wave's 64 (rows of the MFMA result)
    auto chan = [&](int j, int e) { return j * 16 + (lane >> 4) * 4 + e; };
    float bv[NF][NE];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int e = 0; e < NE; ++e) bv[j][e] = p.bias ? p.bias[n0 + wn * 64 + chan(j, e)] : 0.f;
    float ssum[NF][NE], ssq[NF][NE];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int e = 0; e < NE; ++e) ssum[j][e] = ssq[j][e] = 0.f;
    const bool want_stats = p.stat_sum != nullptr;
    float* red = (float*)(smem + RED);
    // everything issued so far (weight tiles 0 and 1, the first patch, the bias row) is waited for here, once
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp();                                   // 1: set-up done, first operands landed (this wave's)

    int pbuf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * WD_BM;
        acc_t acc[NF][NF];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[i][j][e] = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            // what the patch pieces issued during this chunk are for
            const bool last_c = c + 1 == nchunk;
            const int nt = last_c ? tile + (int)gridDim.x : tile;
            const bool nvalid = nt < ntiles;
            const int nm0 = nt * WD_BM, nh0 = tile_h0(nm0), nc0 = last_c ? 0 : (c + 1) * 64;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // everything but the copies issued during the previous step has landed: the previous step issued 3
                // copies per wave when its tap was <= 6 (weight pieces + one patch piece), else 2.  (The epilogue's
                // stores are younger still: the first wait of a tile also waits for most of them.)
                if (tap == 0 || tap == 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                barrier_keep_copies();
                if (tap == 0 && c == 0) stamp();           // 2, 5, ...: the tile's first operands have landed
                auto issue = [&]() {
                    const int t2 = tap + 2;
                    const int tapn = t2 >= 9 ? t2 - 9 : t2;
                    int cn = t2 >= 9 ? c + 1 : c;
                    if (cn == nchunk) cn = 0;                        // the next tile starts over (same weights)
                    stage_b((tap + 2) % 3, tapn * p.C + cn * 64);
                    if (tap <= 6) stage_piece(tap, pbuf ^ 1, nm0, nh0, nc0, nvalid);
                };
                if (grp == 0) issue();
                const int dh = tap / 3, dw = tap % 3;
                const int boff = (tap % 3) * B_BYTES;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    bf16x8 xa[NF], wb[NF];
#pragma unroll
                    for (int i = 0; i < NF; ++i) {
                        int a = a_addr[i][dw];
                        if (ks) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(a) : "v"(a_addr[i][dw]));     // stays in the loop
                        xa[i] = *(const bf16x8*)(smem + a + dh * W2 * PT_ROWB);
                    }
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        int b = b_addr[j];
                        if (ks) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(b) : "v"(b_addr[j]));
                        wb[j] = *(const bf16x8*)(smem + b + boff);
                    }
#pragma unroll
                    for (int i = 0; i < NF; ++i)
#pragma unroll
                        for (int j = 0; j < NF; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
                }
                if (grp == 1) issue();
            }
            // the buffers swap: the next chunk (or tile) reads the patch that was just prefetched
            pbuf ^= 1;
            const int delta = pbuf ? Q_BYTES : -Q_BYTES;
#pragma unroll
            for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int d = 0; d < 3; ++d) a_addr[i][d] += delta;
        }

        stamp();                                   // 3, 6, ...: K loop done
        // ---- epilogue: statistics in registers, bf16 tile through the consumed patch buffer (pbuf ^ 1 after the swap)
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    // the bias joins after the sum over k, as in the 4-wave kernel: the two kernels give the same bits,
                    // so which of them a launch picks never changes a result
                    const float v = acc[i][j][e] + bv[j][e];
                    acc[i][j][e] = v;
                    ssum[j][e] += v;
                    ssq[j][e] += v * v;
                }
        const unsigned stg = (unsigned)(size_t)(lptr_t)smem + (unsigned)((pbuf ^ 1) * Q_BYTES);     // LDS byte address
        barrier_keep_copies();                       // every wave has read the last tap of that patch
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if ((wm >> 1) == pass) {
#pragma unroll
                for (int i = 0; i < NF; ++i) {
                    const int ml = (wm & 1) * 64 + i * F + (lane & (F - 1));
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int e = 0; e < NE; e += 4)      // registers e .. e+3: four consecutive channels
                            lds_store_b64(stg + ml * OROW + (wn * 64 + chan(j, e)) * 2, pack2bf(acc[i][j][e], acc[i][j][e + 1]),
                                          pack2bf(acc[i][j][e + 2], acc[i][j][e + 3]));
                }
            }
            barrier_keep_copies();
            {
                static_assert(128 * (WD_BN / 8) / 512 == 4, "four 16-byte pieces per thread and half-tile");
                u32x4 v[4];
                const unsigned a = stg + (tid >> 4) * OROW + (tid & 15) * 16;
                lds_load4_b128(a, a + 32 * OROW, a + 64 * OROW, a + 96 * OROW, v[0], v[1], v[2], v[3]);
                bf16_t* o = p.out + (long)(m0 + pass * 128 + (tid >> 4)) * p.N + n0 + (tid & 15) * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(u32x4*)(o + (long)q * 32 * p.N) = v[q];
            }
            if (pass == 0) barrier_keep_copies();    // the second half overwrites the staging rows
        }
        stamp();                                   // 4, 7, ...: epilogue issued
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // copies requested past the last tile land before LDS is released
    stamp();                                       // stores and copies complete

    if (want_stats) {
        // sum over the pixels of a fragment = over the 16 consecutive lanes that share a channel
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                ssum[j][e] = pt_row16_sum(ssum[j][e]);
                ssq[j][e] = pt_row16_sum(ssq[j][e]);
            }
        if ((lane & (F - 1)) == 0) {
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int cc = wn * 64 + chan(j, e);
                    red[wm * WD_BN + cc] = ssum[j][e];              // one owner per (wave row, channel) slot
                    red[(4 + wm) * WD_BN + cc] = ssq[j][e];
                }
        }
        barrier_keep_copies();
        if (tid < WD_BN) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[w * WD_BN + tid]; q += red[(4 + w) * WD_BN + tid]; }
            p.stat_sum[(long)blockIdx.x * p.N + n0 + tid] = s;
            p.stat_sq[(long)blockIdx.x * p.N + n0 + tid] = q;
        }
    }
}


// ============================================================================================================
// The weight-stationary form for layers with 64 input channels: 256 pixels x 64 output channels per tile, ONE
// workgroup of 512 threads per CU.
//
// With 64 input channels a tile is only nine K-steps deep, so in the kernels above the per-tile costs -- nine weight
// tiles fetched again for every tile (72 KB against a 43 KB patch), a barrier per K-step, the fill of the copy ring --
// outweigh the matrix work (0.26-0.30 of the matrix-pipe peak at batch 512).  Here the whole [64][9][64] weight block
// of the workgroup's output channels (72 KB) is copied to LDS ONCE and stays; per tile only the patch moves, into
// the other of two patch buffers while the current one is being multiplied, and the nine taps run with no barrier and
// no copy between them: three barriers per tile instead of eleven.  Eight waves of 64 pixels x 32 channels (4 x 2
// fragments: six fragment reads per eight matrix instructions).
// The sum over k runs in the same order as in the other two kernels (tap by tap, two k-steps of 32 each; the bias joins
// after it), so the three give the same bits.
constexpr int WS_BM = 256, WS_BN = 64;

// NW = 8 (the form that is launched): waves of 64 pixels x 32 channels, two per SIMD.  NW = 4: waves of 64 x 64 (eight
// fragment reads per sixteen matrix instructions instead of six per eight: a third less LDS traffic), one per SIMD with
// the whole register file -- measured 10 % slower (0.132 against 0.119 ms for the three launches of a step at batch 512).
template <int W, int NW, bool PIPE>
__global__ __launch_bounds__(64 * NW, NW / 4) void conv3_ws64_kernel(const PatchTN p) {
    constexpr int NT = 64 * NW;                                  // threads
    constexpr int WS_QP = (pt_rows(W, WS_BM) / 8 + NW - 1) / NW; // patch pieces per wave
    constexpr int W2 = W + 2;
    constexpr int QROWS = pt_rows(W, WS_BM);
    constexpr int Q_BYTES = QROWS * PT_ROWB;
    constexpr int QPIECES = QROWS / 8;
    static_assert(QROWS % 8 == 0 && QPIECES <= NW * WS_QP, "patch pieces per wave");
    constexpr int TAP_BYTES = WS_BN * PT_ROWB;                   // one tap's 64 weight rows
    constexpr int WBASE = 2 * Q_BYTES, DUMMY = WBASE + 9 * TAP_BYTES, TOTAL = DUMMY + 1024;
    static_assert(TOTAL <= 163840, "LDS budget");
    constexpr int OROW = WS_BN * 2 + 16;
    static_assert(WS_BM * OROW <= Q_BYTES, "output tile does not fit a patch buffer");
    static_assert(2 * 4 * WS_BN * 4 <= Q_BYTES, "statistics fold does not fit a patch buffer");
    constexpr int NF = 4, NJ = NW == 8 ? 2 : 4, NE = 4;          // pixel fragments, channel fragments, registers per fragment
    constexpr int CW = NJ * 16;                                  // channels per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[TOTAL];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NW == 8 ? wave >> 1 : wave, wn = NW == 8 ? wave & 1 : 0;     // wave tile: pixels [64 wm, +64) x channels [CW wn, +CW)
    const int n0 = blockIdx.y * WS_BN;
    const int ntiles = p.M / WS_BM;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    const unsigned rowb = 128;                                   // 64 input channels
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((unsigned)p.M * rowb), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)((unsigned)p.N * 576u * 2u), 0x00020000);
    auto tile_h0 = [&](int m0t) { return p.G == 1 ? ((m0t >> p.wshift) & (p.H - 1)) : 0; };

    // ---- the weight block: piece q = rows [8 q, +8) of [tap][channel], 72 pieces, 72 / NW per wave
#pragma unroll
    for (int i = 0; i < 72 / NW; ++i) {
        const int q = wave + NW * i;
        const int r = q * 8 + (lane >> 3), tap = r >> 6, n = r & 63;
        const int chunk = (lane & 7) ^ ((n >> 1) & 7);
        const unsigned off = ((unsigned)(n0 + n) * 576u + (unsigned)tap * 64u + chunk * 8) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lptr_t)(smem + WBASE + q * 1024), 16, (int)off, 0, 0, 0);
    }

    // ---- patch pieces of this wave: piece (wave + 8 i) = patch rows [8 (wave + 8 i), +8), lane -> (row, 16-byte chunk)
    int q_st[WS_QP];
#pragma unroll
    for (int i = 0; i < WS_QP; ++i) {
        const int prow = (wave + NW * i) * 8 + (lane >> 3);
        const int per = (p.R + 2) * W2;
        const int g = p.G == 1 ? 0 : (int)(((float)prow + 0.5f) * p.inv_per);
        const int rem = prow - g * per;
        const int rr = (int)(((float)rem + 0.5f) * (1.0f / W2)), xx = rem - rr * W2;
        int st = ((lane & 7) ^ pt_swz(xx)) << 18;
        if (prow < p.PR && xx >= 1 && xx <= W) st |= ((g * p.R + (rr - 1)) * W + (xx - 1) + 256) | (rr << 10) | (1 << 24);
        q_st[i] = st;
    }
    // every wave issues WS_QP copies per patch (a piece past the patch, or a patch that does not exist, is a zero-fill
    // into the dummy kilobyte), so the counted waits below are compile-time constants
    auto stage_piece = [&](int i, int buf, int m0t, int h0t, bool valid) {
        int st = q_st[i];
        asm volatile("" : "+v"(st));
        const int m = m0t + (st & 1023) - 256, r = ((st >> 10) & 255) - 1;
        const bool ok = valid && (st >> 24) && (unsigned)(h0t + r) < (unsigned)p.H && (unsigned)m < (unsigned)p.M;
        const unsigned off = ok ? (unsigned)m * rowb + (unsigned)((st >> 14) & 0x70) : OOB;
        const int piece = wave + NW * i;
        const int dst = piece < QPIECES ? buf * Q_BYTES + piece * 1024 : DUMMY;      // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lptr_t)(smem + dst), 16, (int)off, 0, 0, 0);
    };
    int tile = blockIdx.x;
    {
        const int m0t = tile * WS_BM, h0t = tile_h0(m0t);
#pragma unroll
        for (int i = 0; i < WS_QP; ++i) stage_piece(i, 0, m0t, h0t, true);
    }

    // ---- fragment addresses (patch buffer 0; += / -= Q_BYTES when the buffers swap)
    int a_addr[NF][3];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int ml = wm * 64 + i * 16 + (lane & 15);
        const int g = ml >> p.rwshift, kr = ml & ((1 << p.rwshift) - 1);
        const int r = kr >> p.wshift, x = kr & (W - 1);
        const int prow = (g * (p.R + 2) + r) * W2 + x + 1;            // the pixel one image row up
#pragma unroll
        for (int d = 0; d < 3; ++d)
            a_addr[i][d] = (prow + d - 1) * PT_ROWB + (((lane >> 4) ^ pt_swz(x + d)) << 4);
    }
    int b_addr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int row = wn * CW + j * 16 + (lane & 15);
        b_addr[j] = WBASE + row * PT_ROWB + (((lane >> 4) ^ ((row >> 1) & 7)) << 4);
    }
    auto chan = [&](int j, int e) { return wn * CW + j * 16 + (lane >> 4) * 4 + e; };      // channel of the tile
    float bv[NJ][NE];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < NE; ++e) bv[j][e] = p.bias ? p.bias[n0 + chan(j, e)] : 0.f;
    float ssum[NJ][NE], ssq[NJ][NE];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < NE; ++e) ssum[j][e] = ssq[j][e] = 0.f;
    const bool want_stats = p.stat_sum != nullptr;

    int pbuf = 0;
    bool first = true;
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * WS_BM;
        // this tile's patch (and, the first time, the weights and the bias) has landed; the stores of the previous
        // tile's epilogue (2048 / NT per thread) are younger and may still be in flight
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2048 / NT) : "memory");
        first = false;
        barrier_keep_copies();
        // the next tile's patch goes into the other buffer (last read by the previous epilogue, behind the barrier).
        // A copy holds its wave for 100-200 cycles of issue: the two waves of a SIMD (w and w + 4) request their pieces
        // at opposite ends of the K loop, so that one's copy issue runs under the other's matrix instructions.
        auto stage_next = [&]() {
            const int nt = tile + (int)gridDim.x;
            const int nm0 = nt * WS_BM, nh0 = tile_h0(nm0);
#pragma unroll
            for (int i = 0; i < WS_QP; ++i) stage_piece(i, pbuf ^ 1, nm0, nh0, nt < ntiles);
        };
        const bool copy_first = NW != 8 || wave < 4 || (p.ws64_copy_first != 0);
        if (copy_first) stage_next();
        f32x4 acc[NF][NJ];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // the 18 half-steps (tap, 32-channel half) of the tile.  PIPE: software-pipelined by hand -- the fragments of
        // half-step h + 1 are requested before the matrix instructions of half-step h are issued (the compiler, left
        // alone, emits "six reads, wait, eight matrix instructions" per half-step).  Measured at batch 512: the kernel
        // itself 4 % faster (0.116 -> 0.112 ms for the three launches of a step), the train step 8 us faster on the
        // boxes of the pool that run it in 1.53-1.56 ms and 24-30 us SLOWER on those that run it in 1.57-1.62 ms, with
        // no kernel longer in the trace: the step there follows the energy a kernel draws, not its duration.  The
        // launcher therefore takes the plain form; PIPE stays selectable (tuning key conv3_ws64 = 3).
        bf16x8 xa[2][NF], wb[2][NJ];
        auto fetch = [&](int h, int buf) {
            const int tap = h >> 1, ks = h & 1, dh = tap / 3, dw = tap % 3;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                int a = a_addr[i][dw];
                if (ks) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(a) : "v"(a_addr[i][dw]));
                xa[buf][i] = *(const bf16x8*)(smem + a + dh * W2 * PT_ROWB);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                int b = b_addr[j];
                if (ks) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(b) : "v"(b_addr[j]));
                wb[buf][j] = *(const bf16x8*)(smem + b + tap * TAP_BYTES);
            }
        };
        if (PIPE) {
            fetch(0, 0);
#pragma unroll
            for (int h = 0; h < 18; ++h) {
                if (h + 1 < 18) fetch(h + 1, (h + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);       // the scheduler must not sink the reads back below the matrix instructions
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[h & 1][j], xa[h & 1][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 18; ++h) {
                fetch(h, 0);
#pragma unroll
                for (int i = 0; i < NF; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0][j], xa[0][i], acc[i][j], 0, 0, 0);
            }
        }
        if (!copy_first) stage_next();
        // ---- epilogue: bias and statistics in registers, the bf16 tile through the patch buffer just consumed
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const float v = acc[i][j][e] + bv[j][e];
                    acc[i][j][e] = v;
                    ssum[j][e] += v;
                    ssq[j][e] += v * v;
                }
        const unsigned stg = (unsigned)(size_t)(lptr_t)smem + (unsigned)(pbuf * Q_BYTES);     // LDS byte address
        barrier_keep_copies();                       // every wave has read the last tap of that patch
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int ml = wm * 64 + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                lds_store_b64(stg + ml * OROW + chan(j, 0) * 2, pack2bf(acc[i][j][0], acc[i][j][1]), pack2bf(acc[i][j][2], acc[i][j][3]));
        }
        barrier_keep_copies();
        {
            // 2048 16-byte pieces (256 rows x 8): piece (row, tid & 7), rows tid >> 3 + (NT / 8) * q
            constexpr int RS = NT / 8;                 // rows one pass of the workgroup covers
            u32x4 v[2048 / NT];
            const unsigned a = stg + (tid >> 3) * OROW + (tid & 7) * 16;
#pragma unroll
            for (int h = 0; h < 2048 / NT / 4; ++h)
                lds_load4_b128(a + (4 * h) * RS * OROW, a + (4 * h + 1) * RS * OROW, a + (4 * h + 2) * RS * OROW,
                               a + (4 * h + 3) * RS * OROW, v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]);
            bf16_t* o = p.out + (long)(m0 + (tid >> 3)) * p.N + n0 + (tid & 7) * 8;
#pragma unroll
            for (int q = 0; q < 2048 / NT; ++q) *(u32x4*)(o + (long)q * RS * p.N) = v[q];
        }
        pbuf ^= 1;
        const int delta = pbuf ? Q_BYTES : -Q_BYTES;
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int d = 0; d < 3; ++d) a_addr[i][d] += delta;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // copies requested past the last tile land before LDS is released

    if (want_stats) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                ssum[j][e] = pt_row16_sum(ssum[j][e]);
                ssq[j][e] = pt_row16_sum(ssq[j][e]);
            }
        float* red = (float*)smem;                   // [sum | sum of squares][wave row][channel], in patch buffer 0
        barrier_keep_copies();                       // the last tile's staged rows have been read
        if ((lane & 15) == 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    red[wm * WS_BN + chan(j, e)] = ssum[j][e];
                    red[(4 + wm) * WS_BN + chan(j, e)] = ssq[j][e];
                }
        }
        barrier_keep_copies();
        if (tid < WS_BN) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += red[w * WS_BN + tid]; q += red[(4 + w) * WS_BN + tid]; }
            p.stat_sum[(long)blockIdx.x * p.N + n0 + tid] = s;
            p.stat_sq[(long)blockIdx.x * p.N + n0 + tid] = q;
        }
    }
}

// 0: choose per launch, 1: always the 4-wave kernel, 2: the 8-wave kernel wherever its geometry allows
int g_conv3_wide_mode = 0;
// 0: the weight-stationary kernel for 64-input-channel launches that fill the chip, 1: never, 2: wherever its geometry allows,
// 3: as 0 with the hand-pipelined K loop (PIPE = true; see the kernel), 4: as 0 with every wave requesting the next patch
// at the top of the tile (the A/B partner of the split copy issue)
int g_conv3_ws64_mode = 0;
// 0: choose per launch, 1: never split K inside the workgroup, 2: wherever the chunk count is even
int g_conv3_ksplit_mode = 0;
unsigned long long* g_conv3_dbg = nullptr;      // diagnostic: where the stamping instantiation writes (vae_set_debug_buffer)

static int pt_ilog2(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

template <int W>
static void pt_launch_w(const PatchTN& p, dim3 grid, int bn, int bm, int ks, hipStream_t s) {
    if (ks == 2) {
        if (bn == 64) hipLaunchKernelGGL((conv3_patch_tn_kernel<W, 64, 128, 2>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv3_patch_tn_kernel<W, 128, 128, 2>), grid, dim3(512), 0, s, p);
        return;
    }
    if constexpr (W >= 8) {
        if (bm == 256) { hipLaunchKernelGGL((conv3_patch_tn_kernel<W, 64, 256>), grid, dim3(256), 0, s, p); return; }
    }
    if (bn == 64) hipLaunchKernelGGL((conv3_patch_tn_kernel<W, 64, 128>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv3_patch_tn_kernel<W, 128, 128>), grid, dim3(256), 0, s, p);
}

// tiles of bm pixels hold whole image rows or whole images, and the patch fits the rows reserved for this width
static bool pt_geom_ok(int H, int W, int bm) {
    const long hw = (long)H * W;
    if (bm == 256 && W < 8) return false;
    if (hw < bm && (bm % hw || hw < 16)) return false;
    if (hw >= bm && bm % W) return false;
    const int R = hw >= bm ? bm / W : H, G = hw >= bm ? 1 : (int)(bm / hw);
    return G * (R + 2) * (W + 2) <= pt_rows(W, bm);
}

// 1 when the patch kernel covers the geometry: power-of-two image, 4 <= W <= 64, tiles of 128 pixels that
// hold whole image rows or whole images, 64-channel chunks, N a multiple of 64
bool conv3_patch_tn_ok(int n_img, int H, int W, int Cin, int Cout) {
    if (pt_ilog2(H) < 0 || pt_ilog2(W) < 0 || W < 4 || W > 64) return false;
    if (Cin % 64 || Cout % 64) return false;
    if (!pt_geom_ok(H, W, 128)) return false;
    const long hw = (long)H * W;
    if ((long)n_img * hw * (Cin > Cout ? Cin : Cout) * 2 >= (1L << 32)) return false;
    if ((long)Cout * 9 * Cin >= (1L << 30)) return false;
    return true;
}

// Tiling of a launch: 128-channel tiles where that still gives every CU a workgroup, else 64 channels; with 64
// channels, 256-pixel tiles when those still give every CU two workgroups.  gx workgroups along M each walk
// ceil(tiles / gx) pixel tiles and leave one row of partial statistics.
struct PtPlan {
    int bm, bn, gx, gy, ks;
    bool wide, ws64;
};
static PtPlan pt_plan(long M, int H, int W, int Cin, int Cout) {
    PtPlan q;
    q.wide = false;
    q.ws64 = false;
    q.ks = 1;
    // 64 input channels: the weight-stationary kernel (whole 256-pixel tiles, 16- or 32-wide images: the widths whose
    // two patch buffers fit beside the 72 KB weight block), where its jobs give every CU a workgroup
    if (g_conv3_ws64_mode != 1 && Cin == 64 && (W == 16 || W == 32) && Cout % WS_BN == 0 && M % WS_BM == 0 &&
        pt_geom_ok(H, W, WS_BM)) {
        const long jobs = M / WS_BM * (Cout / WS_BN);
        if (jobs >= 256 || g_conv3_ws64_mode == 2) {
            q.ws64 = true;
            q.bm = WS_BM; q.bn = WS_BN; q.gy = Cout / WS_BN;
            long gx = 256 / q.gy;
            if (gx < 1) gx = 1;
            if (gx > M / WS_BM) gx = M / WS_BM;
            q.gx = (int)gx;
            return q;
        }
    }
    // the 8-wave kernel: whole 256-pixel x 128-channel tiles only, one workgroup per CU; worth it when its jobs
    // fill at least three quarters of the chip
    if (g_conv3_wide_mode != 1 && W >= 8 && Cout % WD_BN == 0 && M % WD_BM == 0 && pt_geom_ok(H, W, WD_BM)) {
        const long jobs = M / WD_BM * (Cout / WD_BN);
        if (jobs >= 192 || g_conv3_wide_mode == 2) {
            q.wide = true;
            q.bm = WD_BM; q.bn = WD_BN; q.gy = Cout / WD_BN;
            long gx = 256 / q.gy;
            if (gx > M / WD_BM) gx = M / WD_BM;
            q.gx = (int)gx;
            return q;
        }
    }
    const long t128 = (M + 127) / 128;
    q.bm = 128;
    q.bn = Cout % 128 == 0 && t128 * (Cout / 128) >= 256 ? 128 : 64;
    if (q.bn == 64 && (M + 255) / 256 * (Cout / 64) >= 512 && M % 256 == 0 && pt_geom_ok(H, W, 256)) q.bm = 256;
    const long ntiles = (M + q.bm - 1) / q.bm;
    q.gy = Cout / q.bn;
    long gx = 256 * pt_wgs(W, q.bn, q.bm) / q.gy;      // every CU holds its full share of resident workgroups
    if (gx > ntiles) gx = ntiles;
    if (gx < 1) gx = 1;
    q.gx = (int)gx;
    // at most one four-wave workgroup per CU: split K between two groups of four waves (an even number of 64-channel chunks)
    // (measured at batch 512: the 4x4 layers 20.6 -> 18.1 us; nothing with the 128-channel tile, whose split form is at
    // the register limit, so that one is only taken on request)
    if (q.bm == 128 && g_conv3_ksplit_mode != 1 && (Cin / 64) % 2 == 0 &&
        (((long)q.gx * q.gy <= 256 && q.bn == 64) || g_conv3_ksplit_mode == 2))
        q.ks = 2;
    return q;
}

int conv3_patch_tn_rows(long M, int H, int W, int Cin, int Cout) { return pt_plan(M, H, W, Cin, Cout).gx; }

int conv3_patch_tn_launch(const void* x, const void* w_packed, const float* bias, void* y, float* stat_sum, float* stat_sq,
                          int n_img, int H, int W, int Cin, int Cout, double flops, double bytes, hipStream_t s) {
    PatchTN p = {};
    p.A = (const bf16_t*)x; p.B = (const bf16_t*)w_packed; p.bias = bias; p.out = (bf16_t*)y;
    p.stat_sum = stat_sum; p.stat_sq = stat_sq;
    p.M = n_img * H * W; p.N = Cout; p.C = Cin; p.H = H; p.wshift = pt_ilog2(W);
    const PtPlan plan = pt_plan(p.M, H, W, Cin, Cout);
    const long hw = (long)H * W;
    if (hw >= plan.bm) { p.G = 1; p.R = plan.bm / W; } else { p.G = (int)(plan.bm / hw); p.R = H; }
    p.rwshift = pt_ilog2(p.R * W);
    p.PR = p.G * (p.R + 2) * (W + 2);
    p.inv_per = 1.0f / (float)((p.R + 2) * (W + 2));
    VAE_CHECK(p.PR <= pt_rows(W, plan.bm) && p.rwshift >= 0, "conv3 patch kernel: internal geometry error (PR=%d)", p.PR);
    dim3 grid(plan.gx, plan.gy);
    if (plan.ws64) {
        p.ws64_copy_first = g_conv3_ws64_mode == 4;
        ProfScope prof("conv3_ws64_kernel", flops, bytes, s);
        if (g_conv3_ws64_mode == 3) {         // the hand-pipelined K loop
            if (W == 16) hipLaunchKernelGGL((conv3_ws64_kernel<16, 8, true>), grid, dim3(512), 0, s, p);
            else hipLaunchKernelGGL((conv3_ws64_kernel<32, 8, true>), grid, dim3(512), 0, s, p);
        } else {
            if (W == 16) hipLaunchKernelGGL((conv3_ws64_kernel<16, 8, false>), grid, dim3(512), 0, s, p);
            else hipLaunchKernelGGL((conv3_ws64_kernel<32, 8, false>), grid, dim3(512), 0, s, p);
        }
        VAE_LAUNCH_CHECK("conv3_ws64_kernel");
        return 0;
    }
    if (plan.wide) {
        ProfScope prof("conv3_wide_kernel", flops, bytes, s);
#define WIDE_V(V)                                                                                      \
        switch (W) {                                                                                   \
            case 8: hipLaunchKernelGGL((conv3_wide_kernel<8, V>), grid, dim3(512), 0, s, p); break;    \
            case 16: hipLaunchKernelGGL((conv3_wide_kernel<16, V>), grid, dim3(512), 0, s, p); break;  \
            case 32: hipLaunchKernelGGL((conv3_wide_kernel<32, V>), grid, dim3(512), 0, s, p); break;  \
            default: hipLaunchKernelGGL((conv3_wide_kernel<64, V>), grid, dim3(512), 0, s, p); break;  \
        }
        if (g_conv3_dbg) { p.dbg = g_conv3_dbg; WIDE_V(true) } else { WIDE_V(false) }
#undef WIDE_V
        VAE_LAUNCH_CHECK("conv3_wide_kernel");
        return 0;
    }
    static char name[4][48] = {"conv3_patch_tn_kernel<64>", "conv3_patch_tn_kernel<128>", "conv3_patch_tn_kernel<64,ksplit>",
                               "conv3_patch_tn_kernel<128,ksplit>"};
    ProfScope prof(name[(plan.bn == 128) + 2 * (plan.ks == 2)], flops, bytes, s);
    switch (W) {
        case 4: pt_launch_w<4>(p, grid, plan.bn, plan.bm, plan.ks, s); break;
        case 8: pt_launch_w<8>(p, grid, plan.bn, plan.bm, plan.ks, s); break;
        case 16: pt_launch_w<16>(p, grid, plan.bn, plan.bm, plan.ks, s); break;
        case 32: pt_launch_w<32>(p, grid, plan.bn, plan.bm, plan.ks, s); break;
        default: pt_launch_w<64>(p, grid, plan.bn, plan.bm, plan.ks, s); break;
    }
    VAE_LAUNCH_CHECK("conv3_patch_tn_kernel");
    return 0;
}

}  // namespace vae
