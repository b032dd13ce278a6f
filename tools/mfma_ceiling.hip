// What a bf16 matrix-instruction loop can reach on this chip, as a file: the MFMA-only loop and the LDS-read + MFMA
// loop of the conv3x3 kernels' wave tile (64 x 64 outputs = 16 accumulators of v_mfma_f32_16x16x32_bf16, eight
// operand fragments per sixteen instructions), on RANDOM operands, with the clock the chip holds inside the loop
// (shader cycles by s_memtime over the constant 100 MHz counter s_memrealtime).  Measurement aid: no counterpart in
// the reference.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_ceiling.hip -o tools/mfma_ceiling
//
//   tools/mfma_ceiling > profiles/r03_mfma_ceiling.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Stamp { unsigned long long cyc, rt; };

typedef __attribute__((address_space(3))) void* lptr_t;
// LDS reads the compiler cannot fold away or hoist out of the loop (it does not see through the asm), and does not wait for
__device__ __forceinline__ bf16x8 lds_read128(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// MODE 0: operands stay in registers.  MODE 1: every fragment is re-read from LDS (ds_read_b128, conflict-free rows of
// 128 B with the chunk index XOR-swizzled by the row, the layout of the conv kernels' weight tiles) right before its
// use: reads, wait, matrix instructions -- the order hipcc gives the conv kernels' K loop.  MODE 2: the same reads
// software-pipelined by hand: the fragments of the next k-half are requested before this half's matrix instructions.
// MODE 3: MODE 1 plus a workgroup barrier in front of every K-step (two k-halves), as the conv kernels have for their
// weight ring.  MODE 4: MODE 3 with the first k-half's four ACTIVATION fragments of the next K-step requested before
// that barrier (the patch does not change between taps; only the weight fragments must wait for the barrier).
// MODE 5: MODE 4 plus the second k-half's fragments requested before the first half's matrix instructions.
// SHAPE 0: 16x16x32 (4 x 4 accumulators of 4 registers), 1: 32x32x16 (2 x 2 accumulators of 16 registers)
template <int MODE, int SHAPE>
__global__ __launch_bounds__(512) void loop_kernel(const uint4* __restrict__ src, float* __restrict__ sink, Stamp* __restrict__ stamps,
                                                   int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];      // an "A" tile and a "B" tile: 128 rows x 128 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * 8; i += blockDim.x) ((uint4*)lds)[i] = src[(blockIdx.x * 2048 + i) & 0xFFFF];
    __syncthreads();
    constexpr int F = SHAPE == 0 ? 16 : 32, NF = 64 / F;          // fragment rows; fragments per 64-row wave tile side
    bf16x8 a[2][NF], b[2][NF];
    int a_addr[NF], b_addr[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int row = (wave & 1) * 64 + i * F + (lane % F), chunk = lane / F;
        a_addr[i] = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
        b_addr[i] = 128 * 128 + a_addr[i];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a[ks][i] = *(const bf16x8*)(lds + (a_addr[i] ^ (ks * 64)));
            b[ks][i] = *(const bf16x8*)(lds + (b_addr[i] ^ (ks * 64)));
        }
    }
    f32x4 acc4[4][4];
    f32x16 acc16[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc4[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc16[i][j][e] = 0.f;
    const unsigned base = (unsigned)(size_t)(lptr_t)lds;
    auto fetch = [&](int ks) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            a[ks][i] = lds_read128(base + (a_addr[i] ^ (ks * 64)));
            b[ks][i] = lds_read128(base + (b_addr[i] ^ (ks * 64)));
        }
    };
    auto mfmas = [&](int ks) {
        if (SHAPE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ks][j], a[ks][i], acc4[i][j], 0, 0, 0);
        } else {
            // the same 64 x 64 x 32 of work: 2 x 2 blocks x two k-quarters of 16 (both take this half's fragments)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ks][j], a[ks][i], acc16[i][j], 0, 0, 0);
        }
    };
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 2) { fetch(0); }
    auto fetch_a = [&](int ks) {
#pragma unroll
        for (int i = 0; i < NF; ++i) a[ks][i] = lds_read128(base + (a_addr[i] ^ (ks * 64)));
    };
    auto fetch_b = [&](int ks) {
#pragma unroll
        for (int i = 0; i < NF; ++i) b[ks][i] = lds_read128(base + (b_addr[i] ^ (ks * 64)));
    };
    if (MODE >= 4) fetch_a(0);
    for (int it = 0; it < iters; ++it) {
        // one "K-step" of the conv kernels: 64 channels = two k-halves
        if (MODE >= 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE == 3) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { fetch(ks); lds_wait(); mfmas(ks); }
        } else if (MODE == 4) {
            fetch_b(0); lds_wait(); mfmas(0);
            fetch(1); lds_wait(); mfmas(1);
            fetch_a(0);                              // the next K-step's activation fragments, ahead of its barrier
        } else if (MODE == 5) {
            fetch_b(0); lds_wait();
            fetch(1); __builtin_amdgcn_sched_barrier(0);
            mfmas(0); __builtin_amdgcn_sched_barrier(0);
            lds_wait();
            fetch_a(0); __builtin_amdgcn_sched_barrier(0);
            mfmas(1); __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (MODE == 1) { fetch(ks); lds_wait(); }
            if (MODE == 2) {
                lds_wait();                         // this half's fragments (requested one half ago) have landed
                fetch(ks ^ 1);                      // the next half's travel under this half's matrix instructions
                __builtin_amdgcn_sched_barrier(0);
            }
            mfmas(ks);
            if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc4[i][j][0] + acc4[i][j][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc16[i][j][0] + acc16[i][j][15];
    sink[blockIdx.x * blockDim.x + tid] = s;        // the stamps go to a buffer of their own; no output depends on them
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = Stamp{c1 - c0, r1 - r0};
}

// The same loop for other wave tiles: NA x NB fragments of 16 (pixels x channels), 16x16x32 instructions, two k-halves
// per iteration; PIPE = 0: reads, wait, matrix instructions; 1: the next half's reads ahead of this half's instructions.
// (NA + NB) reads feed NA * NB instructions: 64 x 32 -> 6 per 8, 64 x 64 -> 8 per 16, 128 x 64 -> 12 per 32.
template <int NA, int NB, int PIPE>
__global__ __launch_bounds__(512) void tile_kernel(const uint4* __restrict__ src, float* __restrict__ sink, Stamp* __restrict__ stamps,
                                                   int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 128 * 128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * 8; i += blockDim.x) ((uint4*)lds)[i] = src[(blockIdx.x * 2048 + i) & 0xFFFF];
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(lptr_t)lds;
    unsigned a_addr[NA], b_addr[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int row = ((wave & 1) * 64 + i * 16 + (lane & 15)) & 127;
        a_addr[i] = base + row * 128 + (((lane >> 4) ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = ((wave >> 1) * 32 + j * 16 + (lane & 15)) & 127;
        b_addr[j] = base + 128 * 128 + row * 128 + (((lane >> 4) ^ ((row >> 1) & 7)) << 4);
    }
    bf16x8 a[2][NA], b[2][NB];
    f32x4 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int ks, int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) a[buf][i] = lds_read128(a_addr[i] ^ (ks * 64));
#pragma unroll
        for (int j = 0; j < NB; ++j) b[buf][j] = lds_read128(b_addr[j] ^ (ks * 64));
    };
    auto mfmas = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[buf][j], a[buf][i], acc[i][j], 0, 0, 0);
    };
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (PIPE) fetch(0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (PIPE) {
                lds_wait();
                fetch(ks ^ 1, ks ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(ks);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                fetch(ks, 0);
                lds_wait();
                mfmas(0);
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][3];
    sink[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = Stamp{c1 - c0, r1 - r0};
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int SHAPE>
static void run(const char* name, int waves_per_simd, const uint4* src, float* sink, Stamp* stamps, bool last) {
    const int threads = 256 * waves_per_simd, blocks = 256, iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // warm the clocks: back-to-back launches for about two seconds before the timed ones
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((loop_kernel<MODE, SHAPE>), dim3(blocks), dim3(threads), 0, 0, src, sink, stamps, iters);
    CK(hipDeviceSynchronize());
    std::vector<double> ms;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((loop_kernel<MODE, SHAPE>), dim3(blocks), dim3(threads), 0, 0, src, sink, stamps, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<Stamp> h(blocks * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> ghz, cyc;
    for (int bidx = 0; bidx < blocks; ++bidx)
        for (int w = 0; w < 4 * waves_per_simd; ++w) {
            const Stamp& st = h[bidx * 8 + w];
            if (st.rt) { ghz.push_back((double)st.cyc / (double)st.rt * 0.1); cyc.push_back((double)st.cyc); }
        }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = 2.0 * 64 * 64 * 64 * (double)iters * blocks * 4 * waves_per_simd;     // one 64 x 64 x 64 K-step per wave and iteration
    const double med = ms[ms.size() / 2];
    const double mfma_per_wave = (SHAPE == 0 ? 32.0 : 16.0) * iters;
    printf("  {\"loop\": \"%s\", \"waves_per_simd\": %d, \"tflops\": %.1f, \"frac_of_2500\": %.3f, \"ms\": %.3f, "
           "\"clock_ghz_in_kernel_median\": %.3f, \"cycles_per_mfma_per_simd\": %.2f}%s\n",
           name, waves_per_simd, flop / med / 1e9, flop / med / 1e9 / 2500.0, med, ghz[ghz.size() / 2],
           cyc[cyc.size() / 2] / mfma_per_wave / waves_per_simd, last ? "" : ",");
}

template <int NA, int NB, int PIPE>
static void run_tile(const char* name, const uint4* src, float* sink, Stamp* stamps, bool last) {
    const int threads = 512, blocks = 256, iters = 20000 * 16 / (NA * NB);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL((tile_kernel<NA, NB, PIPE>), dim3(blocks), dim3(threads), 0, 0, src, sink, stamps, iters);
    CK(hipDeviceSynchronize());
    std::vector<double> ms;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((tile_kernel<NA, NB, PIPE>), dim3(blocks), dim3(threads), 0, 0, src, sink, stamps, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<Stamp> h(blocks * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> ghz, cyc;
    for (const Stamp& st : h)
        if (st.rt) { ghz.push_back((double)st.cyc / (double)st.rt * 0.1); cyc.push_back((double)st.cyc); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double flop = 2.0 * (NA * 16) * (NB * 16) * 64 * (double)iters * blocks * 8;
    const double med = ms[ms.size() / 2];
    printf("  {\"loop\": \"%s\", \"waves_per_simd\": 2, \"tflops\": %.1f, \"frac_of_2500\": %.3f, \"ms\": %.3f, "
           "\"clock_ghz_in_kernel_median\": %.3f, \"cycles_per_mfma_per_simd\": %.2f}%s\n",
           name, flop / med / 1e9, flop / med / 1e9 / 2500.0, med, ghz[ghz.size() / 2],
           cyc[cyc.size() / 2] / (2.0 * NA * NB * iters) / 2, last ? "" : ",");
}

int main() {
    uint4* src; float* sink; Stamp* stamps;
    CK(hipMalloc(&src, 65536 * sizeof(uint4)));
    CK(hipMalloc(&sink, 256 * 512 * sizeof(float)));
    CK(hipMalloc(&stamps, 256 * 8 * sizeof(Stamp)));
    CK(hipMemset(stamps, 0, 256 * 8 * sizeof(Stamp)));
    // random bf16 values in [-1, 1): sign, exponent 118..126, 7 random mantissa bits
    std::vector<unsigned short> h(65536 * 8);
    unsigned long long x = 88172645463325252ull;
    for (auto& v : h) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        v = (unsigned short)(((x >> 20) & 0x8000) | ((118 + (x >> 40) % 9) << 7) | ((x >> 8) & 0x7F));
    }
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("{\"what\": \"bf16 MFMA loops on random operands, one workgroup per CU (256 CUs), 64 x 64 wave tiles; peak 2500 TFLOP/s = 1024 SIMDs x 1024 flop/cycle x 2.4 GHz\",\n");
    printf(" \"device\": \"%s\", \"cus\": %d, \"loops\": [\n", prop.name, prop.multiProcessorCount);
    run<0, 0>("mfma only, 16x16x32", 1, src, sink, stamps, false);
    run<0, 0>("mfma only, 16x16x32", 2, src, sink, stamps, false);
    run<1, 0>("8 ds_read_b128, wait, 16 mfma, 16x16x32", 1, src, sink, stamps, false);
    run<1, 0>("8 ds_read_b128, wait, 16 mfma, 16x16x32", 2, src, sink, stamps, false);
    run<2, 0>("8 ds_read_b128 one half ahead of 16 mfma, 16x16x32", 1, src, sink, stamps, false);
    run<2, 0>("8 ds_read_b128 one half ahead of 16 mfma, 16x16x32", 2, src, sink, stamps, false);
    run<3, 0>("barrier per K-step, then 8 reads, wait, 16 mfma per half, 16x16x32", 2, src, sink, stamps, false);
    run<4, 0>("the same, the next K-step's 4 activation fragments requested before its barrier", 2, src, sink, stamps, false);
    run<5, 0>("the same, and the second half's fragments before the first half's mfma", 2, src, sink, stamps, false);
    run<0, 1>("mfma only, 32x32x16", 1, src, sink, stamps, false);
    run<0, 1>("mfma only, 32x32x16", 2, src, sink, stamps, false);
    run<1, 1>("4 ds_read_b128, wait, 8 mfma, 32x32x16", 1, src, sink, stamps, false);
    run<1, 1>("4 ds_read_b128, wait, 8 mfma, 32x32x16", 2, src, sink, stamps, false);
    run<2, 1>("4 ds_read_b128 one half ahead of 8 mfma, 32x32x16", 1, src, sink, stamps, false);
    run<2, 1>("4 ds_read_b128 one half ahead of 8 mfma, 32x32x16", 2, src, sink, stamps, false);
    run_tile<4, 2, 0>("wave tile 64 x 32: 6 ds_read_b128, wait, 8 mfma", src, sink, stamps, false);
    run_tile<4, 2, 1>("wave tile 64 x 32: 6 reads one half ahead of 8 mfma", src, sink, stamps, false);
    run_tile<4, 4, 0>("wave tile 64 x 64: 8 ds_read_b128, wait, 16 mfma", src, sink, stamps, false);
    run_tile<4, 4, 1>("wave tile 64 x 64: 8 reads one half ahead of 16 mfma", src, sink, stamps, false);
    run_tile<8, 4, 0>("wave tile 128 x 64: 12 ds_read_b128, wait, 32 mfma", src, sink, stamps, false);
    run_tile<8, 4, 1>("wave tile 128 x 64: 12 reads one half ahead of 32 mfma", src, sink, stamps, true);
    printf("]}\n");
    return 0;
}
