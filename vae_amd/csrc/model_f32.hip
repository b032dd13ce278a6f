// VAEConv forward / backward in fp32 (vae_model_cfg.precision = 1): the layer order of vae/models/vae.py:89-122 over
// the kernels of f32_path.hip.  Tensors are NCHW fp32 as in the reference, parameters are read in their own
// layouts straight from the flat parameter buffer, gradients are written in those layouts straight into the flat
// gradient buffer -- every one of the 52, including the biases in front of a train-mode BatchNorm, whose gradient is
// rounding noise around zero here as it is in the reference.
#include "f32_path.h"
#include "model_layout.h"

namespace vae_model {

using namespace vae::f32;
using vae::DropCfg;
using vae::make_drop;

namespace {

struct WsF32 {
    long x;
    long ya[4], aa[4], yb[4], ob[4], arg[3];
    long coef;               // 11 x {scale, shift, mean, rstd} x 256 floats
    long z, klp, h, yd[3], ad[3];
    long g[2], dhead, slab, sums, part, dz, dmu, dlv;
    long slab_floats;
    long total_bytes;
};

void make_ws_f32(const Dims& d, WsF32* w) {
    long o = 0;
    auto put = [&](long bytes) { long r = o; o += align_up(bytes, 256); return r; };
    const long npix = (long)d.N * d.S * d.S;
    w->x = put(npix * d.Cin * 4);
    for (int b = 0; b < 4; ++b) {
        const long h = d.S >> b, e = (long)d.N * h * h * ENC_C[b];
        w->ya[b] = put(e * 4); w->aa[b] = put(e * 4); w->yb[b] = put(e * 4);
        w->ob[b] = put((b < 3 ? e / 4 : e) * 4);
        if (b < 3) w->arg[b] = put(e / 4);
    }
    w->coef = put(11L * 4 * 256 * 4);
    w->z = put((long)d.N * d.L * 4);
    w->klp = put((long)vae_reparam_grid((long)d.N * d.L) * 4);
    w->h = put((long)d.N * d.F * 4);
    const int s8 = d.S / 8;
    for (int u = 0; u < 3; ++u) {
        const long ho = (long)s8 << (u + 1), e = (long)d.N * ho * ho * DEC_CO[u];
        w->yd[u] = put(e * 4); w->ad[u] = put(e * 4);
    }
    for (int i = 0; i < 2; ++i) w->g[i] = put(npix * 64 * 4);
    w->dhead = put(npix * d.Cout * 4);
    long s = 0;
    auto need = [&](int mode, int H, int Ci, int Co, int ks) {
        const long v = slab_need(mode, d.N, H, H, Ci, Co, ks);
        if (v > s) s = v;
    };
    // every GEMM the two passes run (one at a time: the scratch is the largest need)
    for (int b = 0; b < 4; ++b) {
        const int Ci = b ? ENC_C[b - 1] : d.Cin, C = ENC_C[b], H = d.S >> b;
        for (int mode : {CONV_FWD, CONV_DGRAD, CONV_WGRAD}) {
            need(mode, H, Ci, C, 3);
            need(mode, H, C, C, 3);
        }
    }
    for (int u = 0; u < 3; ++u)
        for (int mode : {CONVT_FWD, CONVT_DGRAD, CONVT_WGRAD}) need(mode, s8 << u, DEC_CI[u], DEC_CO[u], 1);
    for (int mode : {CONV_FWD, CONV_DGRAD, CONV_WGRAD}) need(mode, d.S, 64, d.Cout, 1);
    for (int mode : {LIN_FWD, LIN_DGRAD, LIN_WGRAD}) {
        need(mode, 1, d.F, d.L, 1);
        need(mode, 1, d.L, d.F, 1);
    }
    w->slab_floats = s;
    w->slab = put((s > 0 ? s : 64) * 4);
    w->sums = put(2 * 256 * 4);
    w->part = put(chan_part_doubles(d.F) * 8);
    w->dz = put((long)d.N * d.L * 4); w->dmu = put((long)d.N * d.L * 4); w->dlv = put((long)d.N * d.L * 4);
    w->total_bytes = o;
}

struct C32 {
    const vae_model_cfg* cfg;
    Dims d;
    const ParamLayout* pl;
    const float* params;
    float* grads;
    float* bn;
    char* w;
    WsF32 ws;
    hipStream_t s;
    const float* P(int i) const { return params + pl->off[i]; }
    float* G(int i) const { return grads + pl->off[i]; }
    float* F(long off) const { return (float*)(w + off); }
    double* part() const { return (double*)(w + ws.part); }
    float* coef(int bi) const { return (float*)(w + ws.coef) + (long)bi * 4 * 256; }
    float* rmean(int bi) const { return bn + pl->bn_off[bi]; }
    float* rvar(int bi) const { return bn + pl->bn_off[bi] + pl->bn_c[bi]; }
};

int run(const C32& c, int mode, const float* a, const float* b, const float* bias, float* out, int H, int Ci, int Co, int ks,
        int epi = EPI_NONE, int accumulate = 0) {
    Gemm g = {};
    g.a = a; g.b = b; g.bias = bias; g.out = out;
    g.n_img = c.d.N; g.H = H; g.W = H; g.Ci = Ci; g.Co = Co; g.ks = ks;
    g.epi = epi; g.accumulate = accumulate;
    g.slab = c.ws.slab_floats > 0 ? c.F(c.ws.slab) : nullptr;
    return gemm(mode, g, c.s);
}

DropCfg no_drop() { return make_drop(0.0, nullptr, 0, 0, 0, 0); }

int stats(const C32& c, const float* y, int bi, int g_idx, int HW) {
    return bn_stats(y, c.d.N, c.pl->bn_c[bi], HW, c.P(g_idx), c.P(g_idx + 1), c.rmean(bi), c.rvar(bi), c.cfg->train, 0.1f, 1e-5f,
                    c.coef(bi), 256, c.part(), c.s);
}

int encoder_forward(const C32& c, const float* x, const void* const* masks, float* mu, float* logvar) {
    const Dims& d = c.d;
    const WsF32& w = c.ws;
    const int train = c.cfg->train;
    CHECK(hipMemcpyAsync(c.F(w.x), x, (size_t)d.N * d.Cin * d.S * d.S * 4, hipMemcpyDeviceToDevice, c.s) == hipSuccess,
          "vae_model_forward (fp32): could not copy the input batch into the workspace");
    const float* in = c.F(w.x);
    for (int b = 0; b < 4; ++b) {
        const int H = d.S >> b, C = ENC_C[b], Ci = b ? ENC_C[b - 1] : d.Cin;
        TRY(run(c, CONV_FWD, in, c.P(8 * b), c.P(8 * b + 1), c.F(w.ya[b]), H, Ci, C, 3));
        TRY(stats(c, c.F(w.ya[b]), 2 * b, 8 * b + 2, H * H));
        TRY(bn_act_fwd(c.F(w.ya[b]), c.coef(2 * b), 256, c.F(w.aa[b]), nullptr, d.N, C, H, H, 0, no_drop(), c.s));
        TRY(run(c, CONV_FWD, c.F(w.aa[b]), c.P(8 * b + 4), c.P(8 * b + 5), c.F(w.yb[b]), H, C, C, 3));
        TRY(stats(c, c.F(w.yb[b]), 2 * b + 1, 8 * b + 6, H * H));
        const DropCfg drop = make_drop(train ? c.cfg->p_dropout : 0.0, masks ? masks[b] : nullptr, c.cfg->seed, c.cfg->step,
                                       VAE_STREAM_DROPOUT0 + b, c.cfg->rank);
        TRY(bn_act_fwd(c.F(w.yb[b]), c.coef(2 * b + 1), 256, c.F(w.ob[b]), b < 3 ? (unsigned char*)(c.w + w.arg[b]) : nullptr, d.N, C,
                       H, H, b < 3, drop, c.s));
        in = c.F(w.ob[b]);
    }
    // torch.flatten of the NCHW tensor is the tensor itself
    TRY(run(c, LIN_FWD, in, c.P(P_FC_MU_W), c.P(P_FC_MU_B), mu, 1, d.F, d.L, 1));
    return run(c, LIN_FWD, in, c.P(P_FC_LV_W), c.P(P_FC_LV_B), logvar, 1, d.F, d.L, 1);
}

int decoder_forward(const C32& c, const float* z, float* recon) {
    const Dims& d = c.d;
    const WsF32& w = c.ws;
    const int s8 = d.S / 8;
    TRY(run(c, LIN_FWD, z, c.P(P_FC_UP_W), c.P(P_FC_UP_B), c.F(w.h), 1, d.L, d.F, 1, EPI_TANH));
    const float* in = c.F(w.h);          // .view(-1, 256, s8, s8)
    for (int u = 0; u < 3; ++u) {
        const int hh = s8 << u, ho = 2 * hh;
        TRY(run(c, CONVT_FWD, in, c.P(P_UP3_W + 4 * u), c.P(P_UP3_B + 4 * u), c.F(w.yd[u]), hh, DEC_CI[u], DEC_CO[u], 1));
        TRY(stats(c, c.F(w.yd[u]), 8 + u, P_BN3_G + 4 * u, ho * ho));
        TRY(bn_act_fwd(c.F(w.yd[u]), c.coef(8 + u), 256, c.F(w.ad[u]), nullptr, d.N, DEC_CO[u], ho, ho, 0, no_drop(), c.s));
        in = c.F(w.ad[u]);
    }
    return run(c, CONV_FWD, in, c.P(P_HEAD_W), c.P(P_HEAD_B), recon, d.S, 64, d.Cout, 1, EPI_SIGMOID);
}

int backward_decoder(const C32& c, const float* d_recon, const float* recon, const float* mu, const float* logvar,
                     const float* eps_in, const float* dmu_ext, const float* dlv_ext, float kl_scale) {
    const Dims& d = c.d;
    const WsF32& w = c.ws;
    const int s8 = d.S / 8;
    float* cur = c.F(w.g[0]);
    float* other = c.F(w.g[1]);
    float* dhead = c.F(w.dhead);
    float* sums = c.F(w.sums);
    // sigmoid + 1x1 head
    TRY(sigmoid_bwd(d_recon, recon, dhead, (long)d.N * d.Cout * d.S * d.S, c.s));
    TRY(run(c, CONV_WGRAD, dhead, c.F(w.ad[2]), nullptr, c.G(P_HEAD_W), d.S, 64, d.Cout, 1));
    TRY(chan_sum(dhead, d.N, d.Cout, d.S * d.S, c.G(P_HEAD_B), c.part(), c.s));
    TRY(run(c, CONV_DGRAD, dhead, c.P(P_HEAD_W), nullptr, cur, d.S, 64, d.Cout, 1));
    for (int u = 2; u >= 0; --u) {
        const int hh = s8 << u, ho = 2 * hh, Ci = DEC_CI[u], Co = DEC_CO[u];
        const int gi = P_BN3_G + 4 * u;
        TRY(bn_act_bwd(cur, c.F(w.yd[u]), c.coef(8 + u), 256, c.P(gi), nullptr, other, c.G(gi), c.G(gi + 1), sums, c.part(), d.N, Co, ho, ho, 0,
                       no_drop(), c.s));
        TRY(chan_sum(other, d.N, Co, ho * ho, c.G(P_UP3_B + 4 * u), c.part(), c.s));
        const float* xin = u == 0 ? c.F(w.h) : c.F(w.ad[u - 1]);
        TRY(run(c, CONVT_WGRAD, xin, other, nullptr, c.G(P_UP3_W + 4 * u), hh, Ci, Co, 1));
        TRY(run(c, CONVT_DGRAD, other, c.P(P_UP3_W + 4 * u), nullptr, cur, hh, Ci, Co, 1));
    }
    // cur = d tanh(fc_up(z)) [N][F]
    TRY(tanh_bwd(cur, c.F(w.h), other, (long)d.N * d.F, c.s));
    TRY(chan_sum(other, d.N, d.F, 1, c.G(P_FC_UP_B), c.part(), c.s));
    TRY(run(c, LIN_WGRAD, other, c.F(w.z), nullptr, c.G(P_FC_UP_W), 1, d.L, d.F, 1));
    TRY(run(c, LIN_DGRAD, other, c.P(P_FC_UP_W), nullptr, c.F(w.dz), 1, d.L, d.F, 1));
    float* dmu = c.F(w.dmu);
    float* dlv = c.F(w.dlv);
    TRY(vae_reparam_kl_bwd(c.F(w.dz), mu, logvar, eps_in, dmu_ext, dlv_ext, kl_scale, dmu, dlv, nullptr, (long)d.N * d.L, d.L,
                           c.cfg->seed, c.cfg->step, c.cfg->rank, c.s));
    TRY(chan_sum(dmu, d.N, d.L, 1, c.G(P_FC_MU_B), c.part(), c.s));
    TRY(chan_sum(dlv, d.N, d.L, 1, c.G(P_FC_LV_B), c.part(), c.s));
    const float* flat = c.F(w.ob[3]);
    TRY(run(c, LIN_WGRAD, dmu, flat, nullptr, c.G(P_FC_MU_W), 1, d.F, d.L, 1));
    TRY(run(c, LIN_WGRAD, dlv, flat, nullptr, c.G(P_FC_LV_W), 1, d.F, d.L, 1));
    // d(encoder output) -> gradient buffer 0
    TRY(run(c, LIN_DGRAD, dmu, c.P(P_FC_MU_W), nullptr, c.F(w.g[0]), 1, d.F, d.L, 1));
    return run(c, LIN_DGRAD, dlv, c.P(P_FC_LV_W), nullptr, c.F(w.g[0]), 1, d.F, d.L, 1, EPI_NONE, 1);
}

// encoder block b: d(block output) in gradient buffer 0 -> d(block input) in gradient buffer 0
int backward_block(const C32& c, int b, const void* const* masks) {
    const Dims& d = c.d;
    const WsF32& w = c.ws;
    const int H = d.S >> b, C = ENC_C[b];
    float* ga = c.F(w.g[0]);
    float* gb = c.F(w.g[1]);
    float* sums = c.F(w.sums);
    const DropCfg drop = make_drop(c.cfg->p_dropout, masks ? masks[b] : nullptr, c.cfg->seed, c.cfg->step, VAE_STREAM_DROPOUT0 + b,
                                   c.cfg->rank);
    TRY(bn_act_bwd(ga, c.F(w.yb[b]), c.coef(2 * b + 1), 256, c.P(8 * b + 6), b < 3 ? (const unsigned char*)(c.w + w.arg[b]) : nullptr,
                   gb, c.G(8 * b + 6), c.G(8 * b + 7), sums, c.part(), d.N, C, H, H, b < 3, drop, c.s));
    TRY(chan_sum(gb, d.N, C, H * H, c.G(8 * b + 5), c.part(), c.s));
    TRY(run(c, CONV_WGRAD, gb, c.F(w.aa[b]), nullptr, c.G(8 * b + 4), H, C, C, 3));
    TRY(run(c, CONV_DGRAD, gb, c.P(8 * b + 4), nullptr, ga, H, C, C, 3));
    TRY(bn_act_bwd(ga, c.F(w.ya[b]), c.coef(2 * b), 256, c.P(8 * b + 2), nullptr, gb, c.G(8 * b + 2), c.G(8 * b + 3), sums, c.part(), d.N, C, H,
                   H, 0, no_drop(), c.s));
    TRY(chan_sum(gb, d.N, C, H * H, c.G(8 * b + 1), c.part(), c.s));
    const int Ci = b ? ENC_C[b - 1] : d.Cin;
    const float* xin = b ? c.F(w.ob[b - 1]) : c.F(w.x);
    TRY(run(c, CONV_WGRAD, gb, xin, nullptr, c.G(8 * b), H, Ci, C, 3));
    if (b == 0) return 0;          // no gradient with respect to the images
    return run(c, CONV_DGRAD, gb, c.P(8 * b), nullptr, ga, H, Ci, C, 3);
}

}  // namespace

long f32_workspace_bytes(const Dims& d) {
    WsF32 w;
    make_ws_f32(d, &w);
    return w.total_bytes;
}
long f32_klp_offset(const Dims& d) {
    WsF32 w;
    make_ws_f32(d, &w);
    return w.klp;
}

int f32_forward(const vae_model_cfg* cfg, const Dims& d, const ParamLayout& pl, const float* params, float* bn_stats,
                void* workspace, const float* x, const float* eps_in, const void* const* keep_masks, const float* z_in,
                float* recon, float* mu, float* logvar, int parts, void* stream) {
    C32 c;
    c.cfg = cfg; c.d = d; c.pl = &pl; c.params = params; c.grads = nullptr; c.bn = bn_stats; c.w = (char*)workspace;
    c.s = (hipStream_t)stream;
    make_ws_f32(d, &c.ws);
    if (parts & 1) TRY(encoder_forward(c, x, keep_masks, mu, logvar));
    if (!(parts & 2)) return 0;
    const float* z = z_in;
    if (!z) {
        TRY(vae_reparam_kl_fwd(mu, logvar, eps_in, c.F(c.ws.z), nullptr, nullptr, c.F(c.ws.klp), (long)d.N * d.L, cfg->seed, cfg->step,
                               cfg->rank, stream));
        z = c.F(c.ws.z);
    }
    return decoder_forward(c, z, recon);
}

int f32_backward(const vae_model_cfg* cfg, const Dims& d, const ParamLayout& pl, const float* params, void* workspace,
                 const float* d_recon, const float* recon, const float* mu, const float* logvar, const float* eps_in,
                 const void* const* keep_masks, const float* dmu_ext, const float* dlv_ext, float kl_scale, float* grads,
                 int stage, void* stream) {
    C32 c;
    c.cfg = cfg; c.d = d; c.pl = &pl; c.params = params; c.grads = grads; c.bn = nullptr; c.w = (char*)workspace;
    c.s = (hipStream_t)stream;
    make_ws_f32(d, &c.ws);
    if (stage == -1 || stage == 0) TRY(backward_decoder(c, d_recon, recon, mu, logvar, eps_in, dmu_ext, dlv_ext, kl_scale));
    if (stage == -1 || stage == 1) {
        TRY(backward_block(c, 3, keep_masks));
        TRY(backward_block(c, 2, keep_masks));
    }
    if (stage == -1 || stage == 2) {
        TRY(backward_block(c, 1, keep_masks));
        TRY(backward_block(c, 0, keep_masks));
    }
    return 0;
}

}  // namespace vae_model
