// Whole-model entry points: VAEConv.forward / backward as ONE C call each, enqueuing every
// kernel of the path on the caller's stream (no per-layer host round trips, no allocation).
// Layer order follows the reference module exactly:
//   encode  vae/models/vae.py:89-98   (4 x DoubleConv, 3 x MaxPool2d, fc_mu | fc_logvar)
//   reparameterize                     vae/models/vae.py:108-111
//   decode  vae/models/vae.py:114-122 (fc_up + tanh, 3 x ConvTranspose2d + BN + ReLU, 1x1 head + sigmoid)
// Parameters live in ONE flat fp32 buffer in state_dict order (reference shapes), gradients
// in a second flat buffer of the same layout; BatchNorm running stats in a third.
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>

#include "../../include/vae_hip.h"

#include "model_layout.h"

namespace vae { extern int g_conv1_bwd_mode; extern int g_pool_ymax_mode; extern int g_convt_bwd_mode; extern int g_head_sums_mode; int g_fold_defer_mode = 0; int g_slab_reparam_mode = 0; int g_wgrad_batch_mode = 0; int g_conv1_twice_mode = 0; }

namespace {
using namespace vae_model;

// ------------------------------------------------------------------ packed-weight layout (bf16 unless noted)
struct PackLayout {
    long c1a;               // [64][16][8]
    long conv_f[7], conv_d[7];   // c1b, c2a, c2b, c3a, c3b, c4a, c4b
    long fc_f, fc_d, up_f, up_d; // fc_mu|fc_logvar and fc_up
    long ct_f[3], ct_d[3];
    long bcat, bup;         // fp32: [2L], [F] in NHWC order
    long total_bytes;
};
const int CONV_CI[7] = {64, 64, 128, 128, 256, 256, 256};
const int CONV_CO[7] = {64, 128, 128, 256, 256, 256, 256};
const int CONV_PARAM[7] = {4, 8, 12, 16, 20, 24, 28};   // weight index in the param table

void make_pack_layout(const Dims& d, PackLayout* k) {
    long o = 0;
    auto put = [&](long bytes) { long r = o; o += align_up(bytes, 256); return r; };
    k->c1a = put(64 * 128 * 2);
    for (int i = 0; i < 7; ++i) {
        k->conv_f[i] = put((long)CONV_CO[i] * 9 * CONV_CI[i] * 2);
        k->conv_d[i] = put((long)CONV_CO[i] * 9 * CONV_CI[i] * 2);
    }
    k->fc_f = put(2L * d.L * d.F * 2); k->fc_d = put(2L * d.L * d.F * 2);
    k->up_f = put((long)d.L * d.F * 2); k->up_d = put((long)d.L * d.F * 2);
    for (int u = 0; u < 3; ++u) {
        k->ct_f[u] = put((long)DEC_CI[u] * DEC_CO[u] * 4 * 2);
        k->ct_d[u] = put((long)DEC_CI[u] * DEC_CO[u] * 4 * 2);
    }
    k->bcat = put(2L * d.L * 4);
    k->bup = put((long)d.F * 4);
    k->total_bytes = o;
}

// ------------------------------------------------------------------ workspace layout
struct WsLayout {
    long x8;
    long ya[4], aa[4], yb[4], ob[4];
    long coef;              // 11 x 4 x 256 floats: scale, shift, mean, rstd
    long stat_sum, stat_sq; // GEMM epilogue partials
    long zb, klp, h, yd[3], ad[2];
    long bnpart, headpart, headsums, slab, dz, dmu, dlv, dcat, dpre, colp;
    long slab_floats;
    long g[3];              // gradient ping-pong buffers (three: a weight-gradient kernel may still read one)
    long relu_bits;         // first conv in two passes: one bit per element of its activation, "the ReLU passed"
    long route[3];          // pooled blocks: one byte per pooled element, which window element won and whether it is live
    long ymax[3];           // pooled blocks: the conv output at each window's maximum (bf16), for the backward sums
    long dyb[4], dya[4];    // gradients of the two conv outputs of each encoder block: kept until the stage's ONE weight-gradient launch
    long slab2, slab2_floats;
    long total_bytes;
};

// the first conv's BatchNorm backward and weight gradient as one kernel (it holds nine taps x Cin columns in 32)
static bool conv1_one_pass(const Dims& d) { return vae::g_conv1_bwd_mode == 0 && d.Cin <= 3; }
// The first conv's output (128 bytes per pixel made from 16) is never stored: the forward convolves twice -- statistics,
// then BatchNorm + ReLU in the epilogue, leaving the activation and one ReLU bit per element -- and the one-pass
// backward rebuilds y per tile.  Needs that backward form (the separate passes read the stored tensor).
static bool conv1_twice(const Dims& d) {
    return vae::g_conv1_twice_mode == 0 && conv1_one_pass(d) && vae_conv3x3_c8_twice_ok(d.N, d.S, d.S, ENC_C[0]);
}

long enc_elems(const Dims& d, int b) { const long h = d.S >> b; return (long)d.N * h * h * ENC_C[b]; }

int rows_blocks(long M) { return (int)((M + vae_gemm_rows_per_block() - 1) / vae_gemm_rows_per_block()); }

long slab_need(int M, int N, int taps, long K) { return (long)vae_wgrad_ksplit(M, N, taps, K) * taps * M * N; }

void make_ws_layout(const Dims& d, WsLayout* w) {
    long o = 0;
    auto put = [&](long bytes) { long r = o; o += align_up(bytes, 256); return r; };
    const long npix = (long)d.N * d.S * d.S;
    w->x8 = put(npix * 8 * 2);
    w->relu_bits = put(npix * (ENC_C[0] / 8));
    for (int b = 0; b < 4; ++b) {
        const long e = enc_elems(d, b);
        w->ya[b] = put(e * 2); w->aa[b] = put(e * 2); w->yb[b] = put(e * 2);
        w->ob[b] = put((b < 3 ? e / 4 : e) * 2);
        if (b < 3) { w->route[b] = put(e / 4); w->ymax[b] = put(e / 4 * 2); }
    }
    w->coef = put(11L * 4 * 256 * 4);
    const long stat = (long)rows_blocks(npix) * 64;       // widest: first block (64 ch at full res) == up1 (256 cols at half res)
    const long stat2 = (long)rows_blocks(npix / 4) * 256;
    const long statn = (stat > stat2 ? stat : stat2) + VAE_STAT_SCRATCH_ROWS * 256L;   // + the fold's scratch rows
    w->stat_sum = put(statn * 4); w->stat_sq = put(statn * 4);
    w->zb = put((long)d.N * d.L * 2);
    w->klp = put((long)vae_reparam_grid((long)d.N * d.L) * 4);
    w->h = put((long)d.N * d.F * 2);
    const int s8 = d.S / 8;
    for (int u = 0; u < 3; ++u) {
        const long hh = (long)s8 << (u + 1);
        w->yd[u] = put((long)d.N * hh * hh * DEC_CO[u] * 2);
        if (u < 2) w->ad[u] = put((long)d.N * hh * hh * DEC_CO[u] * 2);
    }
    for (int i = 0; i < 3; ++i) w->g[i] = put(npix * 64 * 2);
    for (int b = 0; b < 4; ++b) { w->dyb[b] = put(enc_elems(d, b) * 2); w->dya[b] = put(enc_elems(d, b) * 2); }
    w->bnpart = put(1024L * 512 * 4);
    w->headpart = put((long)vae_head_bwd_blocks(d.N, d.S, d.S) * (3 + d.Cout) * 64 * 4);
    w->headsums = put(7 * 64 * 4);
    // every weight-gradient GEMM of a backward stage keeps its own split-K slabs until the stage's single fold launch:
    // the buffer holds all of them side by side (a few hundred MB of 288 GB)
    auto up64 = [](long v) { return (v + 63) / 64 * 64; };
    long s = up64(slab_need(64, 64, 2, npix));
    if (d.Cin <= 3) { const long v = up64(vae_bn_bwd_conv1_scratch_floats(d.N, d.S, d.S)); if (v > s) s = v; }   // either form of the first conv's backward
    // K splits of the batched weight-gradient launches (one per backward stage: convs of blocks 3 + 2, of blocks 1 + 0)
    int batched_ns[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int stage = 0; stage < 2; ++stage) {
        vae_wgrad_job jobs[VAE_MAX_WGRAD_JOBS];
        int idx[VAE_MAX_WGRAD_JOBS], nj = 0;
        for (int i = 0; i < 7; ++i) {
            const int b = (i + 1) / 2;
            const int h = d.S >> b;
            if ((b >= 2) != (stage == 0) || !vae_conv3_wgrad_patch_ok(h, h, CONV_CI[i], CONV_CO[i])) continue;
            jobs[nj] = vae_wgrad_job{nullptr, nullptr, nullptr, d.N, h, h, CONV_CI[i], CONV_CO[i], 0};
            idx[nj++] = i;
        }
        int ns[VAE_MAX_WGRAD_JOBS];
        if (nj > 0 && vae_conv3_wgrad_multi_plan(jobs, nj, 256, ns) == 0)
            for (int k = 0; k < nj; ++k) batched_ns[idx[k]] = ns[k];
    }
    for (int i = 0; i < 7; ++i) {
        const int b = (i + 1) / 2;     // encoder block of conv i (c1b -> 0, c2a,c2b -> 1, ...)
        const long h = d.S >> b;
        long v = slab_need(CONV_CO[i], CONV_CI[i], 9, (long)d.N * h * h);
        if (vae_conv3_wgrad_patch_ok((int)h, (int)h, CONV_CI[i], CONV_CO[i])) {
            // either form of the launch: one layer at a time, or the stage's layers together
            const long vp = (long)vae_conv3_wgrad_patch_ksplit(d.N, (int)h, (int)h, CONV_CI[i], CONV_CO[i]) * 9 * CONV_CO[i] * CONV_CI[i];
            if (vp > v) v = vp;
            const long vb = (long)batched_ns[i] * 9 * CONV_CO[i] * CONV_CI[i];
            if (vb > v) v = vb;
        }
        s += up64(v);
    }
    for (int u = 0; u < 3; ++u) {
        const long hh = (long)s8 << u;
        long v = slab_need(DEC_CO[u], DEC_CI[u], 4, (long)d.N * hh * hh);
        if (DEC_CO[u] == 64 && DEC_CI[u] == 64) {       // either form of its backward
            const long vf = (long)vae_convt2x2_bwd_fused_slabs(d.N, (int)hh, (int)hh) * 4 * 64 * 64;
            if (vf > v) v = vf;
        }
        s += up64(v);
    }
    s += up64(slab_need(2 * d.L, d.F, 1, d.N));
    s += up64(slab_need(d.F, d.L, 1, d.N));
    w->slab_floats = s;
    w->slab = put(s * 4);
    // split-K slabs of the batch-row nn.Linear layers (main stream; the big slab belongs to the weight-gradient stream)
    long s2 = (long)vae_linear_splitk(d.N, 2 * d.L, d.F) * d.N * 2 * d.L;
    { const long v = (long)vae_linear_splitk(d.N, d.L, d.F) * d.N * d.L; if (v > s2) s2 = v; }
    w->slab2_floats = s2;
    w->slab2 = put(s2 * 4);
    w->dz = put((long)d.N * d.L * 4); w->dmu = put((long)d.N * d.L * 4); w->dlv = put((long)d.N * d.L * 4);
    w->dcat = put((long)d.N * 2 * d.L * 2);
    w->dpre = put((long)d.N * d.F * 2);
    w->colp = put(64L * d.F * 4);
    w->total_bytes = o;
}

struct Ctx {
    Dims d;
    ParamLayout pl;
    PackLayout pk;
    WsLayout ws;
    const vae_model_cfg* cfg;
    float* params;
    float* grads;
    float* bn;
    char* wpack;
    char* w;
    void* stream;
    float* P(int i) const { return params + pl.off[i]; }
    float* G(int i) const { return grads + pl.off[i]; }
    float* coef(int bn_idx, int which) const { return (float*)(w + ws.coef) + ((long)bn_idx * 4 + which) * 256; }
    void* at(long off) const { return w + off; }
    void* pk_at(long off) const { return wpack + off; }
};

int make_ctx(Ctx* c, const vae_model_cfg* cfg, const char* who) {
    TRY(make_dims(cfg, &c->d, who));
    make_param_layout(c->d, &c->pl);
    make_pack_layout(c->d, &c->pk);
    make_ws_layout(c->d, &c->ws);
    c->cfg = cfg;
    return 0;
}

// pack / permute jobs collected on the host stack and issued as ONE launch
struct JobList {
    vae_pack_job j[VAE_MAX_PACK_JOBS];
    int n = 0;
    void add(const float* src, void* dst, int f32, long d0, long d1, long d2, long d3, long s0, long s1, long s2, long s3,
             long base) {
        vae_pack_job& q = j[n++];
        q.src = src; q.dst = dst; q.f32 = f32; q.base = base;
        q.dims[0] = d0; q.dims[1] = d1; q.dims[2] = d2; q.dims[3] = d3;
        q.strides[0] = s0; q.strides[1] = s1; q.strides[2] = s2; q.strides[3] = s3;
    }
};
// fold `ns` slabs: q walks the slab's own dims (d0..d3, row-major), results land at
// out[i0*s0 + i1*s1 + i2*s2 + i3*s3]; (l0..l3) bound the valid (non-padding) index range
int reduce4(const float* slab, int ns, long stride, long slab_base, float* out, long d0, long d1, long d2, long d3, long s0,
            long s1, long s2, long s3, void* stream, long l1 = -1, long l2 = -1) {
    const long dims[4] = {d0, d1, d2, d3}, st[4] = {s0, s1, s2, s3};
    const long lim[4] = {d0, l1 < 0 ? d1 : l1, l2 < 0 ? d2 : l2, d3};
    return vae_reduce_slabs_scatter(slab, ns, stride, slab_base, out, dims, st, lim, 0, 0, stream);
}

// BatchNorm finalize for BN `bi` from the GEMM epilogue partials (train) or running stats (eval)
int bn_finalize(const Ctx& c, int bi, int g_idx, int rows, int fold, double count) {
    const int C = c.pl.bn_c[bi];
    float* rm = c.bn + c.pl.bn_off[bi];
    const int train = c.cfg->train;
    return vae_bn_finalize((float*)c.at(c.ws.stat_sum), (float*)c.at(c.ws.stat_sq), rows, C, fold, count,
                           c.P(g_idx), c.P(g_idx + 1), rm, rm + C, 0.1f, 1e-5f, train ? 0 : 1, c.coef(bi, 0),
                           c.coef(bi, 1), c.coef(bi, 2), c.coef(bi, 3), c.stream);
}

// nn.Linear with an fp32 output and M = batch rows: split-K through the (idle) wgrad slab when that fills the chip
// *slabs_left (optional): when the GEMM is split over K, its slabs (c.ws.slab2) are left for the caller's next kernel to
// add up and *slabs_left is their number; 0: out (/ out2) are written here
int linear_f32(const Ctx& c, const void* a, const void* wgt, const float* bias, float* out, float* out2, int M, int N, int K,
               int epilogue, int* slabs_left = nullptr) {
    if (slabs_left) *slabs_left = 0;
    const int ns = vae_linear_splitk(M, N, K);
    if (ns > 1 && (long)ns * M * N <= c.ws.slab2_floats) {
        if (slabs_left && vae::g_slab_reparam_mode == 0) {
            *slabs_left = ns;
            return vae_linear_gemm_splitk(a, wgt, bias, nullptr, nullptr, (float*)c.at(c.ws.slab2), ns, M, N, K, epilogue, c.stream);
        }
        return vae_linear_gemm_splitk(a, wgt, bias, out, out2, (float*)c.at(c.ws.slab2), ns, M, N, K, epilogue, c.stream);
    }
    return vae_linear_gemm(a, wgt, bias, out, out2, nullptr, nullptr, M, N, K, epilogue, c.stream);
}

}  // namespace

// ==================================================================== public API
extern "C" long vae_model_param_count(const vae_model_cfg* cfg) {
    Ctx c;
    if (make_ctx(&c, cfg, "vae_model_param_count")) return -1;
    return c.pl.total;
}
extern "C" long vae_model_param_offset(const vae_model_cfg* cfg, int index) {
    Ctx c;
    if (make_ctx(&c, cfg, "vae_model_param_offset")) return -1;
    if (index < 0 || index >= P_COUNT) return -1;
    return c.pl.off[index];
}
extern "C" long vae_model_bn_count(const vae_model_cfg* cfg) {
    Ctx c;
    if (make_ctx(&c, cfg, "vae_model_bn_count")) return -1;
    return c.pl.bn_total;
}
extern "C" long vae_model_wpack_bytes(const vae_model_cfg* cfg) {
    Ctx c;
    if (make_ctx(&c, cfg, "vae_model_wpack_bytes")) return -1;
    return c.pk.total_bytes;
}
extern "C" long vae_model_workspace_bytes(const vae_model_cfg* cfg) {
    Ctx c;
    if (make_ctx(&c, cfg, "vae_model_workspace_bytes")) return -1;
    if (cfg->precision == VAE_PRECISION_FP32) return f32_workspace_bytes(c.d);
    return c.ws.total_bytes;
}

extern "C" int vae_model_pack_weights(const vae_model_cfg* cfg, const float* params, void* wpack, void* stream) {
    Ctx c;
    TRY(make_ctx(&c, cfg, "vae_model_pack_weights"));
    CHECK(params && wpack, "vae_model_pack_weights: null pointer");
    c.params = (float*)params; c.wpack = (char*)wpack; c.stream = stream;
    const Dims& d = c.d;
    JobList jl;
    // first conv: [64][16 taps][8 ch]; taps >= 9 / channels >= Cin stay zero (wpack is zero-filled once by the caller)
    jl.add(c.P(0), c.pk_at(c.pk.c1a), 0, 64, d.Cin, 9, 1, 128, 1, 8, 0, 0);
    for (int i = 0; i < 7; ++i) {
        const long Ci = CONV_CI[i], Co = CONV_CO[i];
        jl.add(c.P(CONV_PARAM[i]), c.pk_at(c.pk.conv_f[i]), 0, Co, Ci, 9, 1, 9 * Ci, 1, Ci, 0, 0);
        jl.add(c.P(CONV_PARAM[i]), c.pk_at(c.pk.conv_d[i]), 0, Co, Ci, 9, 1, 1, 9 * Co, -Co, 0, 8 * Co);
    }
    const long L = d.L, F = d.F, hw = d.hw;
    for (int t = 0; t < 2; ++t) {   // fc_mu, fc_logvar -> one [2L][F] operand, columns in NHWC order
        const float* src = c.P(t ? P_FC_LV_W : P_FC_MU_W);
        jl.add(src, c.pk_at(c.pk.fc_f), 0, L, 256, hw, 1, F, 1, 256, 0, t * L * F);
        jl.add(src, c.pk_at(c.pk.fc_d), 0, L, 256, hw, 1, 1, 2 * L, 256 * 2 * L, 0, t * L);
        jl.add(c.P(t ? P_FC_LV_B : P_FC_MU_B), c.pk_at(c.pk.bcat), 1, L, 1, 1, 1, 1, 0, 0, 0, t * L);
    }
    jl.add(c.P(P_FC_UP_W), c.pk_at(c.pk.up_f), 0, 256, hw, L, 1, L, 256 * L, 1, 0, 0);
    jl.add(c.P(P_FC_UP_W), c.pk_at(c.pk.up_d), 0, 256, hw, L, 1, 1, 256, F, 0, 0);
    jl.add(c.P(P_FC_UP_B), c.pk_at(c.pk.bup), 1, 256, hw, 1, 1, 1, 256, 0, 0, 0);
    for (int u = 0; u < 3; ++u) {
        const long Ci = DEC_CI[u], Co = DEC_CO[u];
        const float* src = c.P(P_UP3_W + 4 * u);
        jl.add(src, c.pk_at(c.pk.ct_f[u]), 0, Ci, Co, 4, 1, 1, Ci, Co * Ci, 0, 0);
        jl.add(src, c.pk_at(c.pk.ct_d[u]), 0, Ci, Co, 4, 1, 4 * Co, 1, Co, 0, 0);
    }
    return vae_multi_pack(jl.j, jl.n, stream);
}

// Decoder forward from z (bf16 [N][L] in the workspace) to recon; shared by forward and decode.
static int decoder_forward(const Ctx& c, float* recon) {
    const Dims& d = c.d;
    const WsLayout& w = c.ws;
    const int s8 = d.S / 8;
    const int train = c.cfg->train;
    TRY(vae_linear_gemm(c.at(w.zb), c.pk_at(c.pk.up_f), (const float*)c.pk_at(c.pk.bup), c.at(w.h), nullptr, nullptr,
                        nullptr, d.N, d.F, d.L, VAE_EPI_TANH_BF16, c.stream));
    const void* in = c.at(w.h);
    for (int u = 0; u < 3; ++u) {
        const int hh = s8 << u;   // input grid of this ConvTranspose
        const long Min = (long)d.N * hh * hh;
        float* ss = train ? (float*)c.at(w.stat_sum) : nullptr;
        float* sq = train ? (float*)c.at(w.stat_sq) : nullptr;
        TRY(vae_convt2x2_fwd_gemm(in, c.pk_at(c.pk.ct_f[u]), c.P(P_UP3_B + 4 * u), c.at(w.yd[u]), ss, sq, d.N, hh, hh,
                                  DEC_CI[u], DEC_CO[u], c.stream));
        TRY(bn_finalize(c, 8 + u, P_BN3_G + 4 * u, vae_convt2x2_stat_rows(d.N, hh, hh, DEC_CI[u], DEC_CO[u]),
                        vae_convt2x2_stat_fold(d.N, hh, hh, DEC_CI[u], DEC_CO[u]), 4.0 * (double)Min));
        if (u < 2) {
            TRY(vae_bn_act_fwd(c.at(w.yd[u]), c.coef(8 + u, 0), c.coef(8 + u, 1), c.at(w.ad[u]), d.N, 2 * hh, 2 * hh,
                               DEC_CO[u], 0, 0.0, nullptr, 0, 0, 0, 0, c.stream));
            in = c.at(w.ad[u]);
        }
    }
    return vae_head_fwd(c.at(w.yd[2]), c.coef(10, 0), c.coef(10, 1), c.P(P_HEAD_W), c.P(P_HEAD_B), recon, d.N, d.S, d.S,
                        d.Cout, c.stream);
}

static int encoder_forward(const Ctx& c, const float* x, const void* const* masks, float* mu, float* logvar, int nbn,
                           int* slabs_left = nullptr) {
    const Dims& d = c.d;
    const WsLayout& w = c.ws;
    const int train = c.cfg->train;
    const double p = train ? c.cfg->p_dropout : 0.0;
    float* ss = train ? (float*)c.at(w.stat_sum) : nullptr;
    float* sq = train ? (float*)c.at(w.stat_sq) : nullptr;
    // (num_batches_tracked of the nbn BatchNorm layers this pass runs in train mode)
    TRY(vae_nchw_to_nhwc8_count(x, c.at(w.x8), d.N, d.Cin, d.S, d.S, train ? c.cfg->batches_tracked : nullptr, nbn, c.stream));
    const void* in = c.at(w.x8);
    for (int b = 0; b < 4; ++b) {
        const int H = d.S >> b, C = ENC_C[b];
        const long M = (long)d.N * H * H;
        const int rows_a = b == 0 ? vae_conv3x3_c8_stat_rows(d.N, H, H, C) : vae_conv3x3_stat_rows(d.N, H, H, ENC_C[b - 1], C);
        const int rows = vae_conv3x3_stat_rows(d.N, H, H, C, C);
        const bool twice = b == 0 && conv1_twice(d);
        if (twice) {
            if (train) TRY(vae_conv3x3_c8_stats(in, c.pk_at(c.pk.c1a), c.P(1), ss, sq, d.N, H, H, C, c.stream));
        } else if (b == 0) {
            TRY(vae_conv3x3_c8_gemm(in, c.pk_at(c.pk.c1a), c.P(1), c.at(w.ya[0]), ss, sq, d.N, H, H, C, c.stream));
        } else {
            TRY(vae_conv3x3_gemm(in, c.pk_at(c.pk.conv_f[2 * b - 1]), c.P(8 * b + 1), c.at(w.ya[b]), ss, sq, d.N, H, H,
                                 ENC_C[b - 1], C, c.stream));
        }
        TRY(bn_finalize(c, 2 * b, 8 * b + 2, rows_a, 1, (double)M));
        if (twice)
            TRY(vae_conv3x3_c8_bn_act(in, c.pk_at(c.pk.c1a), c.P(1), c.coef(0, 0), c.coef(0, 1), c.at(w.aa[0]),
                                      train ? c.at(w.relu_bits) : nullptr, d.N, H, H, C, c.stream));
        else
            TRY(vae_bn_act_fwd(c.at(w.ya[b]), c.coef(2 * b, 0), c.coef(2 * b, 1), c.at(w.aa[b]), d.N, H, H, C, 0, 0.0, nullptr,
                               0, 0, 0, 0, c.stream));
        TRY(vae_conv3x3_gemm(c.at(w.aa[b]), c.pk_at(c.pk.conv_f[2 * b]), c.P(8 * b + 5), c.at(w.yb[b]), ss, sq, d.N, H, H, C,
                             C, c.stream));
        TRY(bn_finalize(c, 2 * b + 1, 8 * b + 6, rows, 1, (double)M));
        TRY(vae_bn_act_fwd_route_y(c.at(w.yb[b]), c.coef(2 * b + 1, 0), c.coef(2 * b + 1, 1), c.at(w.ob[b]),
                                 train && b < 3 ? c.at(w.route[b]) : nullptr, train && b < 3 && vae::g_pool_ymax_mode == 0 ? c.at(w.ymax[b]) : nullptr,
                                 d.N, H, H, C, b < 3, p,
                                 masks ? masks[b] : nullptr, c.cfg->seed, c.cfg->step, VAE_STREAM_DROPOUT0 + b, c.cfg->rank,
                                 c.stream));
        in = c.at(w.ob[b]);
    }
    return linear_f32(c, in, c.pk_at(c.pk.fc_f), (const float*)c.pk_at(c.pk.bcat), mu, logvar, d.N, 2 * d.L, d.F,
                      VAE_EPI_F32_SPLIT, slabs_left);
}

extern "C" int vae_model_forward(const vae_model_cfg* cfg, const float* params, float* bn_stats, const void* wpack,
                                 void* workspace, const float* x, const float* eps_in, const void* const* keep_masks,
                                 float* recon, float* mu, float* logvar, void* stream) {
    Ctx c;
    TRY(make_ctx(&c, cfg, "vae_model_forward"));
    if (cfg->precision == VAE_PRECISION_FP32) {
        CHECK(params && bn_stats && workspace && x && recon && mu && logvar, "vae_model_forward: null pointer");
        return f32_forward(cfg, c.d, c.pl, params, bn_stats, workspace, x, eps_in, keep_masks, nullptr, recon, mu, logvar, 3, stream);
    }
    CHECK(params && bn_stats && wpack && workspace && x && recon && mu && logvar, "vae_model_forward: null pointer");
    c.params = (float*)params; c.bn = bn_stats; c.wpack = (char*)wpack; c.w = (char*)workspace; c.stream = stream;
    int ns = 0;      // split-K slabs of the fc_mu | fc_logvar GEMM that the reparameterisation kernel adds up itself
    TRY(encoder_forward(c, x, keep_masks, mu, logvar, 11, &ns));
    if (ns > 0)
        TRY(vae_reparam_kl_fwd_slabs((const float*)c.at(c.ws.slab2), ns, (const float*)c.pk_at(c.pk.bcat), mu, logvar, eps_in,
                                     nullptr, c.at(c.ws.zb), nullptr, (float*)c.at(c.ws.klp), (long)c.d.N * c.d.L, c.d.L,
                                     cfg->seed, cfg->step, cfg->rank, stream));
    else
        TRY(vae_reparam_kl_fwd(mu, logvar, eps_in, nullptr, c.at(c.ws.zb), nullptr, (float*)c.at(c.ws.klp),
                               (long)c.d.N * c.d.L, cfg->seed, cfg->step, cfg->rank, stream));
    return decoder_forward(c, recon);
}

extern "C" int vae_model_encode(const vae_model_cfg* cfg, const float* params, float* bn_stats, const void* wpack,
                                void* workspace, const float* x, const void* const* keep_masks, float* mu,
                                float* logvar, void* stream) {
    Ctx c;
    TRY(make_ctx(&c, cfg, "vae_model_encode"));
    if (cfg->precision == VAE_PRECISION_FP32) {
        CHECK(params && bn_stats && workspace && x && mu && logvar, "vae_model_encode: null pointer");
        return f32_forward(cfg, c.d, c.pl, params, bn_stats, workspace, x, nullptr, keep_masks, nullptr, nullptr, mu, logvar, 1, stream);
    }
    CHECK(params && bn_stats && wpack && workspace && x && mu && logvar, "vae_model_encode: null pointer");
    c.params = (float*)params; c.bn = bn_stats; c.wpack = (char*)wpack; c.w = (char*)workspace; c.stream = stream;
    return encoder_forward(c, x, keep_masks, mu, logvar, 8);
}

extern "C" int vae_model_decode(const vae_model_cfg* cfg, const float* params, float* bn_stats, const void* wpack,
                                void* workspace, const float* z, float* recon, void* stream) {
    Ctx c;
    TRY(make_ctx(&c, cfg, "vae_model_decode"));
    if (cfg->precision == VAE_PRECISION_FP32) {
        CHECK(params && bn_stats && workspace && z && recon, "vae_model_decode: null pointer");
        return f32_forward(cfg, c.d, c.pl, params, bn_stats, workspace, nullptr, nullptr, nullptr, z, recon, nullptr, nullptr, 2, stream);
    }
    CHECK(params && bn_stats && wpack && workspace && z && recon, "vae_model_decode: null pointer");
    c.params = (float*)params; c.bn = bn_stats; c.wpack = (char*)wpack; c.w = (char*)workspace; c.stream = stream;
    TRY(vae_cast_bf16(z, c.at(c.ws.zb), (long)c.d.N * c.d.L, stream));
    return decoder_forward(c, recon);
}

extern "C" int vae_model_kl_partials(const vae_model_cfg* cfg, void* workspace, const float** kl_partial, int* n) {
    Ctx c;
    TRY(make_ctx(&c, cfg, "vae_model_kl_partials"));
    CHECK(workspace && kl_partial && n, "vae_model_kl_partials: null pointer");
    *kl_partial = (const float*)((char*)workspace + (cfg->precision == VAE_PRECISION_FP32 ? f32_klp_offset(c.d) : c.ws.klp));
    *n = vae_reparam_grid((long)c.d.N * c.d.L);
    return 0;
}

// ------------------------------------------------------------------ backward
// Weight gradients are off the critical path of the backward pass: dgrad(L) -> BatchNorm backward(L-1) -> dgrad(L-1)
// ... never reads them, only the optimizer does.  With a second stream (cfg->side_stream) every wgrad kernel and its
// slab fold runs there, beside the main chain, which has many small latency-bound kernels that leave CUs idle.
// Ordering is by events: the side stream waits for the main stream's producer of dY (fork), the main stream waits
// for a wgrad before it overwrites a buffer that wgrad still reads (three gradient buffers rotate, so that wait is
// one to two kernels behind), and every call ends with the main stream waiting for the side stream (join).
struct Fork {
    hipStream_t main, side;
    bool two;
    hipEvent_t* ev;
    int n;
    // slab folds wait for the end of the backward stage: one launch folds them all (each is a few microseconds of
    // work that only the optimizer needs); every weight-gradient GEMM of the stage gets its own piece of the slab buffer
    vae_reduce_job jobs[VAE_MAX_REDUCE_JOBS];
    int njobs;
    long slab_used;      // floats
    bool defer;          // whole backward pass in one call: every fold waits for ONE launch at its end
    // conv3x3 weight gradients wait for the end of the backward stage too: ONE launch computes them all (its
    // workgroups dealt out over the layers), so the stage writes one chip-full of split-K slabs instead of one per layer
    vae_wgrad_job wj[VAE_MAX_WGRAD_JOBS];
    float* wj_out[VAE_MAX_WGRAD_JOBS];
    int nwj;
};

static const int N_EVENTS = 64;
// One pool per device, created on first use under a lock (host threads may drive different devices, or
// the same device on different streams, at once).  Two concurrent backward passes with side streams on
// ONE device would share the pool's events; callers that do that must serialise the calls themselves.
static hipEvent_t* event_pool() {
    static hipEvent_t pool[16][N_EVENTS];
    static bool made[16] = {false};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!made[dev]) {
        for (int i = 0; i < N_EVENTS; ++i)
            if (hipEventCreateWithFlags(&pool[dev][i], hipEventDisableTiming) != hipSuccess) return nullptr;
        made[dev] = true;
    }
    return pool[dev];
}

static int fork_init(Fork* f, void* main, void* side) {
    f->main = (hipStream_t)main;
    f->two = side != nullptr && side != main;
    f->side = f->two ? (hipStream_t)side : (hipStream_t)main;
    f->n = 0;
    f->njobs = 0;
    f->slab_used = 0;
    f->defer = false;
    f->nwj = 0;
    f->ev = f->two ? event_pool() : nullptr;
    CHECK(!f->two || f->ev, "vae_model_backward: could not create the events that order the two streams");
    return 0;
}
// the side stream may start once everything enqueued on the main stream so far has run
static int side_after_main(Fork* f) {
    if (!f->two) return 0;
    hipEvent_t e = f->ev[f->n++ % N_EVENTS];
    CHECK(hipEventRecord(e, f->main) == hipSuccess && hipStreamWaitEvent(f->side, e, 0) == hipSuccess, "stream fork failed");
    return 0;
}
// marks the side stream's progress; main_after() makes the main stream wait for it
static int side_mark(Fork* f, hipEvent_t* out) {
    *out = nullptr;
    if (!f->two) return 0;
    hipEvent_t e = f->ev[f->n++ % N_EVENTS];
    CHECK(hipEventRecord(e, f->side) == hipSuccess, "stream mark failed");
    *out = e;
    return 0;
}
static int main_after(Fork* f, hipEvent_t e) {
    if (!f->two || !e) return 0;
    CHECK(hipStreamWaitEvent(f->main, e, 0) == hipSuccess, "stream join failed");
    return 0;
}

// the next `floats` of the slab buffer
static int slab_take(const Ctx& c, Fork* f, long floats, float** out) {
    CHECK(f->slab_used + floats <= c.ws.slab_floats, "vae_model_backward: split-K slab buffer exhausted (internal sizing error)");
    *out = (float*)c.at(c.ws.slab) + f->slab_used;
    f->slab_used += (floats + 63) / 64 * 64;
    return 0;
}
// record a fold (arguments as reduce4) for the stage's fold launch
static int fold_later(Fork* f, const float* slab, int ns, long stride, long slab_base, float* out, long d0, long d1, long d2,
                      long d3, long s0, long s1, long s2, long s3, long l1 = -1, long l2 = -1) {
    CHECK(f->njobs < VAE_MAX_REDUCE_JOBS, "vae_model_backward: too many pending slab folds");
    vae_reduce_job& q = f->jobs[f->njobs++];
    q.slab = slab; q.nsplit = ns; q.slab_stride = stride; q.slab_base = slab_base; q.out = out; q.base = 0; q.accumulate = 0;
    q.dims[0] = d0; q.dims[1] = d1; q.dims[2] = d2; q.dims[3] = d3;
    q.strides[0] = s0; q.strides[1] = s1; q.strides[2] = s2; q.strides[3] = s3;
    q.limits[0] = d0; q.limits[1] = l1 < 0 ? d1 : l1; q.limits[2] = l2 < 0 ? d2 : l2; q.limits[3] = d3;
    return 0;
}
// fold everything recorded so far, on the stream the weight gradients run on; the slab buffer is free again after it
static int flush_folds(Fork* f, bool last = false) {
    if (f->defer && !last) return 0;
    if (f->njobs > 0) TRY(vae_multi_reduce_slabs_scatter(f->jobs, f->njobs, f->side));
    f->njobs = 0;
    f->slab_used = 0;
    return 0;
}

static int wgrad_to(const Ctx& c, Fork* f, float** slab_out, const void* P, const void* Q, int M, int N, long K, int ldp,
                    int ldq, int mode, int H, int W, int taps, int* nsplit_out, void* stream) {
    const int ns = vae_wgrad_ksplit(M, N, taps, K);
    *nsplit_out = ns;
    TRY(slab_take(c, f, (long)ns * taps * M * N, slab_out));
    return vae_wgrad_gemm(P, Q, *slab_out, M, N, K, ldp, ldq, mode, H, W, ns, stream);
}

// 3x3 conv weight gradient into the slab: nine-taps-per-workgroup kernel where it applies
static int conv3_wgrad_to(const Ctx& c, Fork* f, float** slab_out, const void* dy, const void* x, int Co, int Ci, int H,
                          int* nsplit_out, void* stream) {
    if (vae_conv3_wgrad_patch_ok(H, H, Ci, Co)) {
        const int ns = vae_conv3_wgrad_patch_ksplit(c.d.N, H, H, Ci, Co);
        *nsplit_out = ns;
        TRY(slab_take(c, f, (long)ns * 9 * Co * Ci, slab_out));
        return vae_conv3_wgrad_patch(dy, x, *slab_out, c.d.N, H, H, Ci, Co, ns, stream);
    }
    return wgrad_to(c, f, slab_out, dy, x, Co, Ci, (long)c.d.N * H * H, Co, Ci, VAE_WGRAD_CONV3, H, H, 9, nsplit_out, stream);
}

static bool wgrad_batched(int H, int Ci, int Co) { return vae::g_wgrad_batch_mode == 0 && vae_conv3_wgrad_patch_ok(H, H, Ci, Co); }

// Conv2d 3x3 weight gradient of one layer into grad_out ([Co][Ci][3][3]): queued for the stage's single launch when the
// patch kernel covers the geometry (dy must then stay untouched until flush_wgrads), else launched now
static int conv3_wgrad(const Ctx& c, Fork* f, const void* dy, const void* x, int Co, int Ci, int H, float* grad_out, void* stream,
                       bool* queued) {
    *queued = wgrad_batched(H, Ci, Co);
    if (*queued) {
        CHECK(f->nwj < VAE_MAX_WGRAD_JOBS, "vae_model_backward: too many pending weight gradients");
        vae_wgrad_job& q = f->wj[f->nwj];
        q.dy = dy; q.x = x; q.slab = nullptr; q.n_img = c.d.N; q.H = H; q.W = H; q.Cin = Ci; q.Cout = Co; q.nsplit = 0;
        f->wj_out[f->nwj++] = grad_out;
        return 0;
    }
    int ns = 1;
    float* slab = nullptr;
    TRY(conv3_wgrad_to(c, f, &slab, dy, x, Co, Ci, H, &ns, stream));
    // slab [co][tap][ci] -> Conv2d weight [co][ci][tap]
    return fold_later(f, slab, ns, 9L * Co * Ci, 0, grad_out, Co, 9, Ci, 1, 9L * Ci, 1, 9, 0);
}

// the stage's pending weight gradients: plan the K splits, ONE launch, their folds recorded for the stage's fold launch
static int flush_wgrads(const Ctx& c, Fork* f) {
    if (f->nwj == 0) return 0;
    int ns[VAE_MAX_WGRAD_JOBS];
    TRY(vae_conv3_wgrad_multi_plan(f->wj, f->nwj, 256, ns));
    for (int i = 0; i < f->nwj; ++i) {
        vae_wgrad_job& q = f->wj[i];
        q.nsplit = ns[i];
        TRY(slab_take(c, f, (long)ns[i] * 9 * q.Cout * q.Cin, &q.slab));
    }
    TRY(side_after_main(f));
    TRY(vae_conv3_wgrad_patch_multi(f->wj, f->nwj, f->side));
    for (int i = 0; i < f->nwj; ++i) {
        const vae_wgrad_job& q = f->wj[i];
        TRY(fold_later(f, q.slab, q.nsplit, 9L * q.Cout * q.Cin, 0, f->wj_out[i], q.Cout, 9, q.Cin, 1, 9L * q.Cin, 1, 9, 0));
    }
    f->nwj = 0;
    return 0;
}

// stage 0: head .. fc heads (everything after the encoder); leaves d(encoder output) in gradient buffer 0
static int backward_decoder(const Ctx& c, Fork* f, const float* d_recon, const float* recon, const float* mu,
                            const float* logvar, const float* eps_in, const float* dmu_ext, const float* dlv_ext,
                            float kl_scale) {
    const Dims& d = c.d;
    const WsLayout& w = c.ws;
    const int s8 = d.S / 8;
    void* sd = f->side;
    int ns = 1;
    float* slab = nullptr;
    // 1x1 head + bn1
    // [d gamma1 | d beta1] and [d head.weight | d head.bias] are adjacent in the flat layout
    // with the default widths the head's gradient is never stored: its sums (and the logit gradients, 16 bytes per
    // pixel, parked in gradient buffer 0) now, the tensor itself tile by tile inside the last ConvTranspose's backward
    const bool head_fused = vae::g_convt_bwd_mode == 0 && vae::g_head_sums_mode == 0 &&
                            vae_convt2x2_bwd_fused_ok(d.N, s8 << 2, s8 << 2, DEC_CI[2], DEC_CO[2]);
    if (head_fused)
        TRY(vae_head_bwd_sums(d_recon, recon, c.at(w.yd[2]), c.coef(10, 0), c.coef(10, 1), c.coef(10, 2), c.coef(10, 3),
                              c.P(P_HEAD_W), (float*)c.at(w.headpart), c.G(P_BN1_G), c.G(P_HEAD_W), (float*)c.at(w.g[0]), d.N,
                              d.S, d.S, d.Cout, c.stream));
    else
        TRY(vae_head_bwd(d_recon, recon, c.at(w.yd[2]), c.coef(10, 0), c.coef(10, 1), c.coef(10, 2), c.coef(10, 3),
                         c.P(P_HEAD_W), (float*)c.at(w.headpart), c.G(P_BN1_G), c.G(P_HEAD_W), c.at(w.g[0]), d.N, d.S, d.S,
                         d.Cout, c.stream));
    // up1, up2, up3 (u = 2, 1, 0): dy in buffer x -> (side: weight grad) ; input grad -> y ; bn backward -> z
    int x = 0;
    hipEvent_t w_prev = nullptr;       // the wgrad that reads the buffer the next dgrad overwrites
    for (int u = 2; u >= 0; --u) {
        const int y = (x + 1) % 3, z = (x + 2) % 3;
        const int hh = s8 << u;    // input grid
        const long Min = (long)d.N * hh * hh;
        const int Ci = DEC_CI[u], Co = DEC_CO[u];
        const void* xin = u == 0 ? c.at(w.h) : c.at(w.ad[u - 1]);
        if (vae_convt2x2_bwd_fused_ok(d.N, hh, hh, Ci, Co)) {
            // both gradients from one read of dY, on the main stream (there is no separate weight-gradient kernel to move;
            // the side stream's later waits on the main stream order the fold behind it)
            const int nsl = vae_convt2x2_bwd_fused_slabs(d.N, hh, hh);
            TRY(slab_take(c, f, (long)nsl * 4 * Co * Ci, &slab));
            TRY(main_after(f, w_prev));    // buffer y was read by the previous iteration's wgrad
            if (u == 2 && head_fused)
                TRY(vae_head_convt_bwd_fused(c.at(w.yd[2]), (const float*)c.at(w.g[x]), c.coef(10, 0), c.coef(10, 1), c.coef(10, 2),
                                             c.coef(10, 3), c.G(P_BN1_G), c.P(P_HEAD_W), d.Cout, xin, c.pk_at(c.pk.ct_d[u]),
                                             c.at(w.g[y]), slab, d.N, hh, hh, c.stream));
            else
                TRY(vae_convt2x2_bwd_fused(c.at(w.g[x]), xin, c.pk_at(c.pk.ct_d[u]), c.at(w.g[y]), slab, d.N, hh, hh, Ci, Co, c.stream));
            TRY(fold_later(f, slab, nsl, 4L * Co * Ci, 0, c.G(P_UP3_W + 4 * u), Co, 4, Ci, 1, 4, 1, 4L * Co, 0));
            w_prev = nullptr;
        } else {
        TRY(side_after_main(f));
        TRY(wgrad_to(c, f, &slab, c.at(w.g[x]), xin, Co, Ci, Min, Co, Ci, VAE_WGRAD_CONVT, hh, hh, 4, &ns, sd));
        // slab [co][tap][ci] -> ConvTranspose2d weight [ci][co][tap]
        TRY(fold_later(f, slab, ns, 4L * Co * Ci, 0, c.G(P_UP3_W + 4 * u), Co, 4, Ci, 1, 4, 1, 4L * Co, 0));
        hipEvent_t w_this;
        TRY(side_mark(f, &w_this));
        // (the bias in front of a train-mode BatchNorm has an identically zero gradient: its slot is left as it is)
        TRY(main_after(f, w_prev));    // buffer y was read by the previous iteration's wgrad
        TRY(vae_convt2x2_dgrad_gemm(c.at(w.g[x]), c.pk_at(c.pk.ct_d[u]), c.at(w.g[y]), d.N, hh, hh, Ci, Co, c.stream));
        w_prev = w_this;
        }
        if (u > 0) {
            const int bi = 8 + (u - 1);
            TRY(vae_bn_act_bwd(c.at(w.g[y]), c.at(w.yd[u - 1]), c.coef(bi, 0), c.coef(bi, 1), c.coef(bi, 2), c.coef(bi, 3),
                               (float*)c.at(w.bnpart), c.G(P_BN3_G + 4 * (u - 1)), c.at(w.g[z]), d.N, hh, hh, DEC_CO[u - 1], 0,
                               0.0, nullptr, 0, 0, 0, 0, c.stream));
            x = z;
        } else {
            x = y;
        }
    }
    // g[x] = d h (grad of tanh output) [N][F]
    const int nr = vae_tanh_bwd_rows(d.N);
    TRY(vae_tanh_bwd(c.at(w.g[x]), c.at(w.h), c.at(w.dpre), (float*)c.at(w.colp), d.N, d.F, c.stream));
    // fc_up.bias grad: column sums, NHWC order -> parameter order via a permuting reduce
    // (a sum over rows is a fold of `rows` slabs of one row each: it rides in the stage's fold launch)
    TRY(fold_later(f, (float*)c.at(w.colp), nr, d.F, 0, c.G(P_FC_UP_B), 1, d.hw, 256, 1, 0, 1, d.hw, 0));
    TRY(side_after_main(f));
    TRY(wgrad_to(c, f, &slab, c.at(w.dpre), c.at(w.zb), d.F, d.L, d.N, d.F, d.L, VAE_WGRAD_PLAIN, 1, 1, 1, &ns, sd));
    // slab [(p, c)][l] -> fc_up.weight [(c, p)][l]
    TRY(fold_later(f, slab, ns, (long)d.F * d.L, 0, c.G(P_FC_UP_W), d.hw, 256, d.L, 1, d.L, (long)d.hw * d.L, 1, 0));
    int nsz = 0;     // split-K slabs of d z that the reparameterisation backward adds up itself
    TRY(linear_f32(c, c.at(w.dpre), c.pk_at(c.pk.up_d), nullptr, (float*)c.at(w.dz), nullptr, d.N, d.L, d.F, VAE_EPI_F32, &nsz));
    TRY(vae_reparam_kl_bwd_slabs(nsz > 0 ? (const float*)c.at(w.slab2) : (const float*)c.at(w.dz), nsz, mu, logvar, eps_in, dmu_ext, dlv_ext, kl_scale, (float*)c.at(w.dmu),
                           (float*)c.at(w.dlv), c.at(w.dcat), (long)d.N * d.L, d.L, c.cfg->seed, c.cfg->step, c.cfg->rank,
                           c.stream));
    // fc_mu.bias / fc_logvar.bias gradients: column sums of d mu, d logvar
    TRY(fold_later(f, (const float*)c.at(w.dmu), d.N, d.L, 0, c.G(P_FC_MU_B), 1, 1, 1, d.L, 0, 0, 0, 1));
    TRY(fold_later(f, (const float*)c.at(w.dlv), d.N, d.L, 0, c.G(P_FC_LV_B), 1, 1, 1, d.L, 0, 0, 0, 1));
    TRY(side_after_main(f));
    TRY(wgrad_to(c, f, &slab, c.at(w.dcat), c.at(w.ob[3]), 2 * d.L, d.F, d.N, 2 * d.L, d.F, VAE_WGRAD_PLAIN, 1, 1, 1, &ns, sd));
    for (int t = 0; t < 2; ++t)
        // slab rows [t*L, (t+1)*L) x columns (p, c) -> fc weight [l][(c, p)]
        TRY(fold_later(f, slab, ns, 2L * d.L * d.F, (long)t * d.L * d.F, c.G(t ? P_FC_LV_W : P_FC_MU_W), d.L, d.hw, 256, 1,
                       d.F, 1, d.hw, 0));
    TRY(flush_folds(f));
    // d(encoder output) [N][F] bf16 -> buffer 0 (its last reader, the first ConvT wgrad, is behind the events waited above)
    TRY(main_after(f, w_prev));
    return vae_linear_gemm(c.at(w.dcat), c.pk_at(c.pk.fc_d), nullptr, c.at(w.g[0]), nullptr, nullptr, nullptr, d.N, d.F,
                           2 * d.L, VAE_EPI_BF16, c.stream);
}

// encoder block b: expects d(block output) in buffer (3 - b) % 3, leaves d(block input) in the next one.
// *w_carry: the previous block's last wgrad (it reads the buffer this block's first dgrad overwrites).
// The gradients of the block's two conv outputs go to buffers of their own when their weight gradients wait for the
// stage's single launch (they must outlive the block), else into the rotation as before.
static int backward_block(const Ctx& c, Fork* f, int b, const void* const* masks, hipEvent_t* w_carry) {
    const Dims& d = c.d;
    const WsLayout& w = c.ws;
    const int H = d.S >> b, C = ENC_C[b];
    const long M = (long)d.N * H * H;
    const int ia = (3 - b) % 3, ib = (ia + 1) % 3, ic = (ia + 2) % 3;
    void* sd = f->side;
    int ns = 1;
    const double p = c.cfg->p_dropout;
    const int Ci = b ? ENC_C[b - 1] : d.Cin;
    void* dyb = wgrad_batched(H, C, C) ? c.at(w.dyb[b]) : c.at(w.g[ib]);
    void* dya = b > 0 && wgrad_batched(H, Ci, C) ? c.at(w.dya[b]) : c.at(w.g[ia]);
    // second conv + BN + ReLU + dropout (+ pool): a -> dyb
    TRY(vae_bn_act_bwd_route_y(c.at(w.g[ia]), c.at(w.yb[b]), c.coef(2 * b + 1, 0), c.coef(2 * b + 1, 1), c.coef(2 * b + 1, 2),
                             c.coef(2 * b + 1, 3), (float*)c.at(w.bnpart), c.G(8 * b + 6), dyb,
                             b < 3 ? c.at(w.route[b]) : nullptr, b < 3 && vae::g_pool_ymax_mode == 0 ? c.at(w.ymax[b]) : nullptr, d.N, H, H, C, b < 3, p,
                             masks ? masks[b] : nullptr,
                             c.cfg->seed, c.cfg->step, VAE_STREAM_DROPOUT0 + b, c.cfg->rank, c.stream));
    TRY(side_after_main(f));
    bool queued = false;
    TRY(conv3_wgrad(c, f, dyb, c.at(w.aa[b]), C, C, H, c.G(8 * b + 4), sd, &queued));
    hipEvent_t w1 = nullptr;
    if (!queued) TRY(side_mark(f, &w1));
    TRY(main_after(f, *w_carry));       // buffer c was read by the previous block's second wgrad
    TRY(vae_conv3x3_gemm(dyb, c.pk_at(c.pk.conv_d[2 * b]), nullptr, c.at(w.g[ic]), nullptr, nullptr, d.N, H, H, C, C,
                         c.stream));
    // first conv + BN + ReLU: c -> dya
    if (b == 0 && conv1_one_pass(d)) {
        // nothing needs the first conv's input gradient: BatchNorm backward and weight gradient in one pass, no tensor written
        float* scratch = nullptr;
        TRY(slab_take(c, f, vae_bn_bwd_conv1_scratch_floats(d.N, H, H), &scratch));
        if (conv1_twice(d))
            TRY(vae_bn_bwd_conv1_bits(c.at(w.g[ic]), c.at(w.relu_bits), c.at(w.x8), c.P(0), c.P(1), c.coef(0, 0), c.coef(0, 2),
                                      c.coef(0, 3), scratch, c.G(2), c.G(0), d.N, H, H, d.Cin, c.stream));
        else
            TRY(vae_bn_bwd_conv1(c.at(w.g[ic]), c.at(w.ya[0]), c.at(w.x8), c.coef(0, 0), c.coef(0, 1), c.coef(0, 2), c.coef(0, 3),
                                 scratch, c.G(2), c.G(0), d.N, H, H, d.Cin, c.stream));
        TRY(side_after_main(f));           // the stage's fold launch follows it: the scratch rows sit in the slab buffer
        return side_mark(f, w_carry);
    }
    TRY(vae_bn_act_bwd(c.at(w.g[ic]), c.at(w.ya[b]), c.coef(2 * b, 0), c.coef(2 * b, 1), c.coef(2 * b, 2), c.coef(2 * b, 3),
                       (float*)c.at(w.bnpart), c.G(8 * b + 2), dya, d.N, H, H, C, 0, 0.0, nullptr, 0, 0, 0, 0, c.stream));
    TRY(side_after_main(f));
    if (b == 0) {
        float* slab = nullptr;
        TRY(wgrad_to(c, f, &slab, dya, c.at(w.x8), C, 64, M, C, 8, VAE_WGRAD_CONV3_C8, H, H, 2, &ns, sd));
        // slab [co][16 taps][8 ch] (taps >= 9, channels >= Cin are padding) -> weight [co][ci][tap]
        TRY(fold_later(f, slab, ns, 64L * 128, 0, c.G(0), 64, 16, 8, 1, 9L * d.Cin, 1, 9, 0, 9, d.Cin));
        return side_mark(f, w_carry);
    }
    TRY(conv3_wgrad(c, f, dya, c.at(w.ob[b - 1]), C, Ci, H, c.G(8 * b), sd, &queued));
    if (queued) *w_carry = nullptr;
    else TRY(side_mark(f, w_carry));
    TRY(main_after(f, w1));             // buffer b is still read by this block's first wgrad
    return vae_conv3x3_gemm(dya, c.pk_at(c.pk.conv_d[2 * b - 1]), nullptr, c.at(w.g[ib]), nullptr, nullptr, d.N, H, H,
                            C, Ci, c.stream);
}

extern "C" int vae_model_backward(const vae_model_cfg* cfg, const float* params, const void* wpack, void* workspace,
                                  const float* d_recon, const float* recon, const float* mu, const float* logvar,
                                  const float* eps_in, const void* const* keep_masks, const float* dmu_ext,
                                  const float* dlv_ext, float kl_scale, float* grads, int stage, void* stream) {
    Ctx c;
    TRY(make_ctx(&c, cfg, "vae_model_backward"));
    CHECK(params && (wpack || cfg->precision == VAE_PRECISION_FP32) && workspace && d_recon && recon && mu && logvar && grads,
          "vae_model_backward: null pointer");
    CHECK(cfg->train, "vae_model_backward: the forward pass must have run in train mode");
    CHECK(stage >= -1 && stage <= 2, "vae_model_backward: stage=%d out of range (-1 all, 0 decoder+latent, 1 conv4+conv3, 2 conv2+conv1)", stage);
    if (cfg->precision == VAE_PRECISION_FP32)
        return f32_backward(cfg, c.d, c.pl, params, workspace, d_recon, recon, mu, logvar, eps_in, keep_masks, dmu_ext, dlv_ext,
                            kl_scale, grads, stage, stream);
    c.params = (float*)params; c.grads = grads; c.wpack = (char*)wpack; c.w = (char*)workspace; c.stream = stream;
    Fork f;
    TRY(fork_init(&f, stream, cfg->side_stream));
    f.defer = stage == -1 && vae::g_fold_defer_mode == 0;
    hipEvent_t carry = nullptr;
    if (stage == -1 || stage == 0)
        TRY(backward_decoder(c, &f, d_recon, recon, mu, logvar, eps_in, dmu_ext, dlv_ext, kl_scale));
    if (stage == -1 || stage == 1) {
        TRY(backward_block(c, &f, 3, keep_masks, &carry));
        TRY(backward_block(c, &f, 2, keep_masks, &carry));
        TRY(flush_wgrads(c, &f));
        TRY(flush_folds(&f));
    }
    if (stage == -1 || stage == 2) {
        TRY(backward_block(c, &f, 1, keep_masks, &carry));
        TRY(backward_block(c, &f, 0, keep_masks, &carry));
        TRY(flush_wgrads(c, &f));
        TRY(flush_folds(&f, true));
    }
    // join: whatever the caller enqueues next on `stream` (all-reduce of this stage, the optimizer) sees every gradient
    hipEvent_t last;
    TRY(side_mark(&f, &last));
    return main_after(&f, last);
}
