// Parameter / BatchNorm-statistics layout of VAEConv (state_dict order, reference shapes) and the argument checks
// shared by the whole-model entry points of both precisions (model.cpp: bf16 operands; model_f32.cpp: fp32).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/vae_hip.h"

namespace vae {
void set_error(const char* fmt, ...);
}

#define CHECK(cond, ...)                   \
    do {                                   \
        if (!(cond)) {                     \
            ::vae::set_error(__VA_ARGS__); \
            return 1;                      \
        }                                  \
    } while (0)
#define TRY(expr)                 \
    do {                          \
        int e__ = (expr);         \
        if (e__) return e__;      \
    } while (0)

namespace vae_model {

static const int ENC_C[4] = {64, 128, 256, 256};
static const int DEC_CI[3] = {256, 128, 64};   // up3, up2, up1
static const int DEC_CO[3] = {128, 64, 64};

inline long align_up(long v, long a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------ parameter layout
enum {
    P_FC_MU_W = 32, P_FC_MU_B, P_FC_LV_W, P_FC_LV_B, P_FC_UP_W, P_FC_UP_B,
    P_UP3_W, P_UP3_B, P_BN3_G, P_BN3_B, P_UP2_W, P_UP2_B, P_BN2_G, P_BN2_B,
    P_UP1_W, P_UP1_B, P_BN1_G, P_BN1_B, P_HEAD_W, P_HEAD_B, P_COUNT
};
// encoder block b (0..3): conv a weight = 8b, bias 8b+1, gamma 8b+2, beta 8b+3, conv b weight 8b+4, ...

struct Dims {
    int N, Cin, Cout, L, S, hw, F;   // hw = (S/8)^2, F = 256*hw
};

struct ParamLayout {
    long off[P_COUNT], numel[P_COUNT], total;
    long bn_off[11], bn_total;   // running_mean at bn_off, running_var at bn_off + C
    int bn_c[11];
};

inline int make_dims(const vae_model_cfg* c, Dims* d, const char* who) {
    CHECK(c, "%s: null config", who);
    CHECK(c->struct_size == sizeof(vae_model_cfg),
          "%s: vae_model_cfg.struct_size=%u but this library's struct is %zu bytes (ABI revision %d): the caller was built "
          "against another revision of include/vae_hip.h", who, c->struct_size, sizeof(vae_model_cfg), VAE_HIP_VERSION);
    CHECK(c->batch > 0, "%s: batch must be > 0 (got %d)", who, c->batch);
    CHECK(c->in_channels >= 1 && c->in_channels <= 8, "%s: in_channels=%d not supported (1..8)", who, c->in_channels);
    CHECK(c->out_channels >= 1 && c->out_channels <= 4, "%s: out_channels=%d not supported (1..4)", who, c->out_channels);
    CHECK(c->image_size >= 8 && c->image_size % 8 == 0, "%s: image_size=%d must be a positive multiple of 8", who, c->image_size);
    CHECK(c->latent >= 64 && c->latent % 64 == 0, "%s: latent_dims=%d must be a positive multiple of 64", who, c->latent);
    CHECK(c->p_dropout >= 0.0 && c->p_dropout < 1.0, "%s: p_dropout=%f out of [0,1)", who, c->p_dropout);
    CHECK(c->precision == VAE_PRECISION_BF16 || c->precision == VAE_PRECISION_FP32, "%s: precision=%d (0 bf16, 1 fp32)", who,
          c->precision);
    CHECK((long)c->batch * c->image_size * c->image_size < (1L << 31), "%s: batch x image_size^2 exceeds 2^31 pixels", who);
    d->N = c->batch; d->Cin = c->in_channels; d->Cout = c->out_channels; d->L = c->latent; d->S = c->image_size;
    d->hw = (d->S / 8) * (d->S / 8);
    d->F = 256 * d->hw;
    return 0;
}

inline void make_param_layout(const Dims& d, ParamLayout* pl) {
    long o = 0;
    auto put = [&](int idx, long n) {
        pl->off[idx] = o;
        pl->numel[idx] = n;
        o += align_up(n, 4);
    };
    int ci = d.Cin;
    for (int b = 0; b < 4; ++b) {
        const int co = ENC_C[b];
        put(8 * b + 0, (long)co * ci * 9); put(8 * b + 1, co); put(8 * b + 2, co); put(8 * b + 3, co);
        put(8 * b + 4, (long)co * co * 9); put(8 * b + 5, co); put(8 * b + 6, co); put(8 * b + 7, co);
        ci = co;
    }
    put(P_FC_MU_W, (long)d.L * d.F); put(P_FC_MU_B, d.L);
    put(P_FC_LV_W, (long)d.L * d.F); put(P_FC_LV_B, d.L);
    put(P_FC_UP_W, (long)d.F * d.L); put(P_FC_UP_B, d.F);
    for (int u = 0; u < 3; ++u) {
        const int base = P_UP3_W + 4 * u;
        put(base, (long)DEC_CI[u] * DEC_CO[u] * 4); put(base + 1, DEC_CO[u]); put(base + 2, DEC_CO[u]); put(base + 3, DEC_CO[u]);
    }
    put(P_HEAD_W, (long)d.Cout * 64); put(P_HEAD_B, d.Cout);
    pl->total = o;
    long bo = 0;
    for (int i = 0; i < 11; ++i) {
        const int C = i < 8 ? ENC_C[i / 2] : DEC_CO[i - 8];
        pl->bn_c[i] = C;
        pl->bn_off[i] = bo;
        bo += 2 * C;
    }
    pl->bn_total = bo;
}

// ------------------------------------------------------------------ fp32 execution mode (model_f32.hip)
// cfg->precision = 1: the same entry points run NCHW fp32 kernels that read the parameters in their own layouts
// (no packed copies); the workspace has its own layout.
long f32_workspace_bytes(const Dims& d);
long f32_klp_offset(const Dims& d);           // byte offset of the KL partial sums in that workspace
// parts: 1 encoder (x -> mu, logvar), 2 reparameterisation + decoder (mu, logvar -> recon), 3 both;
// z_in (parts = 2 only): decode this z instead of sampling one
int f32_forward(const vae_model_cfg* cfg, const Dims& d, const ParamLayout& pl, const float* params, float* bn_stats,
                void* workspace, const float* x, const float* eps_in, const void* const* keep_masks, const float* z_in,
                float* recon, float* mu, float* logvar, int parts, void* stream);
int f32_backward(const vae_model_cfg* cfg, const Dims& d, const ParamLayout& pl, const float* params, void* workspace,
                 const float* d_recon, const float* recon, const float* mu, const float* logvar, const float* eps_in,
                 const void* const* keep_masks, const float* dmu_ext, const float* dlv_ext, float kl_scale, float* grads,
                 int stage, void* stream);

}  // namespace vae_model
