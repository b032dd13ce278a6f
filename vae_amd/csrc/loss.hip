// Fused reconstruction-loss forward + backward, and the final ELBO combine.
//
// One streaming pass reads recon and target once, writes d(loss)/d(recon) once and
// leaves one fp32 partial sum per workgroup; a single-workgroup kernel then folds the
// partials (in double) into  loss = (recon_sum + kl_weight * kl_sum) / batch.
// HBM-bound: 12 algorithmic bytes per element (two fp32 reads, one fp32 write).
#include "common.h"
#include "elbo.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

template <int KIND>
__device__ __forceinline__ void loss_elem(float r, float t, float gs, float& l, float& g) {
    if (KIND == VAE_RECON_L1) {
        const float d = r - t;
        l = fabsf(d);
        g = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    } else if (KIND == VAE_RECON_MSE) {
        const float d = r - t;
        l = d * d;
        g = 2.f * d * gs;
    } else if (KIND == VAE_RECON_BCE) {
        // F.binary_cross_entropy: -(t max(log r, -100) + (1 - t) max(log(1 - r), -100)) with 1 - r rounded to fp32 first,
        // as ATen forms it.  Logs on the hardware log2 (v_log_f32, 1 ulp) times ln 2: ~1e-7 per term, against the
        // 1e-5 bar on the sum; the accurate logf + log1pf pair made this kernel ALU-bound at a third of the HBM rate.
        const float lr = fmaxf(__logf(r), -100.f);
        const float l1r = fmaxf(__logf(1.f - r), -100.f);
        l = -(t * lr + (1.f - t) * l1r);
        // d/dr = (r - t) / max(r (1 - r), 1e-12)   (ATen binary_cross_entropy_backward); v_rcp_f32 is 1 ulp
        g = (r - t) * __builtin_amdgcn_rcpf(fmaxf(r * (1.f - r), 1e-12f)) * gs;
    } else if (KIND == VAE_RECON_BCE_LOGITS) {
        // binary_cross_entropy_with_logits on logits clamped to [-10, 10]
        const float x = fminf(fmaxf(r, -10.f), 10.f);
        const float e = __expf(-fabsf(x));
        const float inv = __builtin_amdgcn_rcpf(1.f + e);
        l = fmaxf(x, 0.f) - x * t + __logf(1.f + e);          // e in [e^-10, 1]: 1 + e carries e to 2^-24 absolute
        const float sg = x >= 0.f ? inv : e * inv;
        g = (r >= -10.f && r <= 10.f) ? (sg - t) * gs : 0.f;   // clamp passes no gradient outside
    } else {
        // F.binary_cross_entropy_with_logits itself: max(x,0) - x t + log1p(exp(-|x|)), gradient sigmoid(x) - t.
        // log1p(e) = e - e^2/2 + ... for tiny e (1 + e would round it away), hardware log otherwise
        const float e = __expf(-fabsf(r));
        const float inv = __builtin_amdgcn_rcpf(1.f + e);
        l = fmaxf(r, 0.f) - r * t + (e < 1e-4f ? e - 0.5f * e * e : __logf(1.f + e));
        const float sg = r >= 0.f ? inv : e * inv;
        g = (sg - t) * gs;
    }
}

// Streaming shape (measured on 3 x 256 MiB): one workgroup per contiguous run of LOSS_CHUNK elements, two float4 per
// lane and array, non-temporal loads and stores, as many workgroups as there are runs -- 6.3 TB/s.  The same arithmetic
// as a grid-stride loop over a grid capped at 2048 workgroups with plain loads ran at 5.0-5.3 TB/s.
constexpr int LOSS_CHUNK = 2048;        // elements per workgroup = 256 lanes x 2 x float4

typedef __attribute__((ext_vector_type(4))) float f32x4v;
__device__ __forceinline__ float4 ld_stream(const float4* p) {
    const f32x4v v = __builtin_nontemporal_load((const f32x4v*)p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream(float4* p, const float4& g) {
    const f32x4v v = {g.x, g.y, g.z, g.w};
    __builtin_nontemporal_store(v, (f32x4v*)p);
}

// GRAD: 0 no gradient, 1 plain stores (the gradient is read back at once by the next kernel: leave it in L2), 2 non-temporal
// stores (a gradient larger than the caches)
template <int KIND, int GRAD>
__global__ __launch_bounds__(256) void recon_loss_kernel(const float* __restrict__ recon,
                                                         const float* __restrict__ target,
                                                         float* __restrict__ grad,
                                                         float* __restrict__ partial, long n,
                                                         float gs) {
    __shared__ float red[8];
    const long n4 = n >> 2;
    float acc = 0.f;
    const float4* r4 = (const float4*)recon;
    const float4* t4 = (const float4*)target;
    float4* g4 = (float4*)grad;
    const long i = (long)blockIdx.x * (LOSS_CHUNK / 4) + threadIdx.x, j = i + 256;
    const bool one = i < n4, two = j < n4;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f), t = r, r2 = r, t2 = r;
    if (one) { r = ld_stream(r4 + i); t = ld_stream(t4 + i); }
    if (two) { r2 = ld_stream(r4 + j); t2 = ld_stream(t4 + j); }
    if (one) {
        float4 g;
        float l0, l1, l2, l3;
        loss_elem<KIND>(r.x, t.x, gs, l0, g.x);
        loss_elem<KIND>(r.y, t.y, gs, l1, g.y);
        loss_elem<KIND>(r.z, t.z, gs, l2, g.z);
        loss_elem<KIND>(r.w, t.w, gs, l3, g.w);
        acc += (l0 + l1) + (l2 + l3);
        if (GRAD == 2) st_stream(g4 + i, g); else if (GRAD == 1) g4[i] = g;
    }
    if (two) {
        float4 g;
        float l0, l1, l2, l3;
        loss_elem<KIND>(r2.x, t2.x, gs, l0, g.x);
        loss_elem<KIND>(r2.y, t2.y, gs, l1, g.y);
        loss_elem<KIND>(r2.z, t2.z, gs, l2, g.z);
        loss_elem<KIND>(r2.w, t2.w, gs, l3, g.w);
        acc += (l0 + l1) + (l2 + l3);
        if (GRAD == 2) st_stream(g4 + j, g); else if (GRAD == 1) g4[j] = g;
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0) {
        for (long k = (n4 << 2) + threadIdx.x; k < n; k += blockDim.x) {
            float l, g;
            loss_elem<KIND>(recon[k], target[k], gs, l, g);
            acc += l;
            if (GRAD) grad[k] = g;
        }
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[0] = (recon_sum + kl_weight*kl_sum) * inv_batch ; out[1] = recon_sum ; out[2] = kl_sum
__global__ __launch_bounds__(256) void elbo_combine_kernel(const ElboCombine e) {
    __shared__ double red[4];
    elbo_combine_body(e, red);
}

}  // namespace vae

using namespace vae;

extern "C" int vae_recon_loss_grid(long n_elems) {
    long blocks = (n_elems + vae::LOSS_CHUNK - 1) / vae::LOSS_CHUNK;       // one workgroup (and one partial sum) per run
    if (blocks < 1) blocks = 1;
    return blocks > 0x7FFFFFFFL ? -1 : (int)blocks;
}

extern "C" int vae_recon_loss_fwd_bwd(const float* recon, const float* target, float* grad,
                                      float* partial, long n_elems, int kind, float grad_scale,
                                      void* stream) {
    VAE_CHECK(recon && target && partial, "vae_recon_loss_fwd_bwd: null pointer");
    VAE_CHECK(n_elems > 0, "vae_recon_loss_fwd_bwd: n_elems must be > 0 (got %ld)", n_elems);
    VAE_CHECK(kind >= 0 && kind <= 4, "vae_recon_loss_fwd_bwd: unknown kind %d", kind);
    VAE_CHECK(vae_recon_loss_grid(n_elems) > 0, "vae_recon_loss_fwd_bwd: n_elems=%ld too large", n_elems);
    VAE_CHECK(((uintptr_t)recon & 15) == 0 && ((uintptr_t)target & 15) == 0 &&
                  ((uintptr_t)grad & 15) == 0,
              "vae_recon_loss_fwd_bwd: pointers must be 16-byte aligned");
    const int grid = vae_recon_loss_grid(n_elems);
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(grad ? "recon_loss_kernel<fwd+bwd>" : "recon_loss_kernel<fwd>", 0.0, (grad ? 12.0 : 8.0) * n_elems, s);
    const bool big = n_elems >= (16L << 20);      // 64 MB of gradient and up: it will not be found in a cache anyway
#define LAUNCH(K)                                                                              \
    if (grad && big)                                                                           \
        hipLaunchKernelGGL((recon_loss_kernel<K, 2>), dim3(grid), dim3(256), 0, s, recon,      \
                           target, grad, partial, n_elems, grad_scale);                        \
    else if (grad)                                                                             \
        hipLaunchKernelGGL((recon_loss_kernel<K, 1>), dim3(grid), dim3(256), 0, s, recon,      \
                           target, grad, partial, n_elems, grad_scale);                        \
    else                                                                                       \
        hipLaunchKernelGGL((recon_loss_kernel<K, 0>), dim3(grid), dim3(256), 0, s, recon,      \
                           target, grad, partial, n_elems, grad_scale);
    switch (kind) {
        case VAE_RECON_L1: LAUNCH(VAE_RECON_L1); break;
        case VAE_RECON_MSE: LAUNCH(VAE_RECON_MSE); break;
        case VAE_RECON_BCE: LAUNCH(VAE_RECON_BCE); break;
        case VAE_RECON_BCE_LOGITS: LAUNCH(VAE_RECON_BCE_LOGITS); break;
        default: LAUNCH(VAE_RECON_BCE_LOGITS_RAW); break;
    }
#undef LAUNCH
    VAE_LAUNCH_CHECK("recon_loss_kernel");
    return 0;
}

// ---------------------------------------------------------------- allocation-free variant
// The per-workgroup partial sums of vae_recon_loss_fwd_bwd normally live in a buffer the caller allocates.  For callers
// that cannot allocate (code under graph capture, serving loops, harnesses that forbid hipMalloc) the same two kernels
// run with the partials in a fixed array that is part of the library's code object.  Each GPU holds its own copy of the
// array at its own address; the address is looked up once per device (hipGetSymbolAddress) and only that device pointer
// is ever handed to a kernel -- the array's name in host code is a host-side stand-in, not a device address.
// The first call on a device does that lookup (it may load the code object): under HIP graph capture, make one
// call on the device BEFORE the capture begins -- any later call only enqueues two kernels.
#define VAE_STATIC_PARTIALS 16384
static __device__ __align__(256) float scratch[VAE_STATIC_PARTIALS];

#include <mutex>
static int static_partials(float** out) {
    static float* ptr_of[64] = {nullptr};
    static std::mutex mu;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        vae::set_error("vae_recon_loss_fwd_bwd_static: no current device (hipGetDevice failed or index %d >= 64)", dev);
        return 3;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr_of[dev]) {
        float* p = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            vae::set_error("vae_recon_loss_fwd_bwd_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return 3;
        }
        ptr_of[dev] = p;
    }
    *out = ptr_of[dev];
    return 0;
}

extern "C" long vae_recon_loss_static_max_elems(void) { return (long)VAE_STATIC_PARTIALS * vae::LOSS_CHUNK; }

extern "C" int vae_recon_loss_fwd_bwd_static(const float* recon, const float* target, float* grad, float* out3,
                                             long n_elems, int kind, float grad_scale, void* stream) {
    VAE_CHECK(out3, "vae_recon_loss_fwd_bwd_static: null output");
    VAE_CHECK(n_elems <= vae_recon_loss_static_max_elems(),
              "vae_recon_loss_fwd_bwd_static: n_elems=%ld exceeds the maximum of %ld elements this entry point accepts "
              "(use vae_recon_loss_fwd_bwd with a caller-provided partial buffer)", n_elems, vae_recon_loss_static_max_elems());
    const int grid = vae_recon_loss_grid(n_elems);
    VAE_CHECK(grid <= VAE_STATIC_PARTIALS, "vae_recon_loss_fwd_bwd_static: %d partial sums do not fit the fixed array of %d",
              grid, VAE_STATIC_PARTIALS);
    float* partial = nullptr;
    const int rc = static_partials(&partial);
    if (rc) return rc;
    // slots [0, grid) are overwritten by the first kernel and only those are read by the second: nothing to clear
    const int e = vae_recon_loss_fwd_bwd(recon, target, grad, partial, n_elems, kind, grad_scale, stream);
    if (e) return e;
    return vae_elbo_combine(partial, grid, nullptr, 0, 0.f, 1.f, out3, stream);
}

extern "C" int vae_elbo_combine(const float* recon_partial, int n_recon, const float* kl_partial,
                                int n_kl, float kl_weight, float inv_batch, float* out3,
                                void* stream) {
    VAE_CHECK(out3 && (recon_partial || n_recon == 0) && (kl_partial || n_kl == 0),
              "vae_elbo_combine: null pointer");
    const ElboCombine e = {recon_partial, kl_partial, out3, n_recon, n_kl, kl_weight, inv_batch};
    hipLaunchKernelGGL(elbo_combine_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, e);
    VAE_LAUNCH_CHECK("elbo_combine_kernel");
    return 0;
}
