/* libvae_hip.so -- C ABI of the MI355X (gfx950) kernels behind the conv-VAE hot path.
 *
 * The reference (jbthejedi/vae) has no FFI or operator plug-in interface: its hot path is
 * plain torch.nn calls.  Each entry point below therefore names the torch call site in the
 * reference that it replaces (file:line under the reference root).
 *
 * Conventions
 *   - every pointer is a device pointer (tensor.data_ptr()); activations are NHWC bf16 unless
 *     the name says otherwise; "f32" pointers are float.
 *   - the library never allocates, frees or keeps a pointer past return; the caller owns all
 *     outputs and workspaces.  Work is only ENQUEUED on `stream` (a hipStream_t); nothing
 *     synchronises.
 *   - return 0 on success; non-zero on a rejected argument or failed launch, with the reason
 *     in vae_last_error() (thread-local).
 */
#ifndef VAE_HIP_H
#define VAE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ABI revision: bumped whenever a struct layout, an argument list or the meaning of a returned size changes.
 * A binding compares vae_version() with the value it was written for before it makes any other call (vae_amd/_lib.py
 * does); vae_model_cfg carries its own size as a second check that needs no call order. */
#define VAE_HIP_VERSION 3

/* reconstruction-term kinds */
#define VAE_RECON_L1 0          /* F.l1_loss(reduction="sum")                 vae/training/train.py:55        */
#define VAE_RECON_MSE 1         /* F.mse_loss(reduction="sum")                vae/practice/vae_conv.py:160-165 */
#define VAE_RECON_BCE 2         /* F.binary_cross_entropy(reduction="sum")    vae/bak/vae_conv.py:183-190     */
#define VAE_RECON_BCE_LOGITS 3  /* BCE-with-logits on clamp(x,-10,10)         vae/practice/vae_conv.py:146-151 */
#define VAE_RECON_BCE_LOGITS_RAW 4 /* F.binary_cross_entropy_with_logits(reduction="sum") itself, no clamp: what
                                    * vae_loss(..., loss=F.binary_cross_entropy_with_logits) computes, vae/training/train.py:46 */

/* Philox stream ids: one per DoubleConv dropout (vae/models/vae.py:137) + eps (vae.py:110) */
#define VAE_STREAM_DROPOUT0 0
#define VAE_STREAM_EPS 4

/* GEMM epilogues of vae_linear_gemm */
#define VAE_EPI_BF16 0        /* out bf16 [M][N]                                  */
#define VAE_EPI_TANH_BF16 1   /* out bf16 = tanh(acc + bias)                      */
#define VAE_EPI_F32 2         /* out f32 [M][N]                                   */
#define VAE_EPI_F32_SPLIT 3   /* out f32 [M][N/2] (cols < N/2), out2 f32 [M][N/2] */

int vae_version(void);
const char* vae_last_error(void);

/* Optional measurement aid (no counterpart in the reference, which has no profiler): between
 * vae_prof_start and vae_prof_stop every GEMM / norm / loss launch is bracketed by HIP events
 * on its own stream.  vae_prof_stop synchronises the device and returns one row per kernel
 * name: launches, summed milliseconds, and the summed ALGORITHMIC flops / bytes the launchers
 * declared (names: 64 bytes per row).  Returns the number of distinct kernels. */
int vae_prof_start(int max_launches);
int vae_prof_stop(char* names, int* launches, double* total_ms, double* flops, double* bytes,
                  int max_rows);

/* Measurement aid (no counterpart in the reference): picks between kernel variants that compute the same result, so
 * that they can be timed against each other inside one process.  Keys: "conv3_wide" = 0 choose per launch (default),
 * 1 always the 4-wave conv3x3 kernel, 2 the 8-wave kernel wherever its geometry allows; "conv3_ksplit" = 0 default,
 * 1 never split K between two wave groups of a workgroup, 2 wherever the chunk count is even; "stream_gemm" = 0 the
 * streaming kernel for the first conv and for ConvTranspose forwards with large outputs (default), 1 the tiled GEMM
 * always, 2 the streaming kernel wherever its geometry allows (the *_stat_rows functions follow the setting); "wgrad_pc" =
 * 0 the conv3x3 weight gradient with separate copying and computing waves for long K ranges (default), 1 always, 2 never; "conv3_ws64" =
 * 0 the weight-stationary conv3x3 kernel for launches with 64 input channels whose tiles fill the chip (default), 1 never, 2 wherever
 * its geometry allows (16- or 32-wide images, whole 256-pixel tiles), 3 as 0 with a hand-pipelined K loop (faster alone, not in the step on every box),
 * 4 as 0 with every wave requesting the next patch at the top of a tile (default: the two waves of a SIMD at opposite ends of the K loop); "conv1_bwd" = 0 the whole-model backward runs
 * the first conv's BatchNorm backward and weight gradient as vae_bn_bwd_conv1 where Cin <= 3 (default), 1 as separate passes;
 * "pool_ymax" = 0 the whole-model passes keep the conv output at each pooling window's maximum for the backward sums (default), 1 they
 * re-read the windows; "head_sums" = 0 vae_head_bwd forms its per-channel sums on the matrix pipe (default), 1 in vector registers; "fold_defer" = 0
 * vae_model_backward(stage = -1) folds every split-K slab in one launch at its end (default), 1 in one launch per stage;
 * "slab_reparam" = 0 the whole-model passes let the reparameterisation kernels add up the split-K slabs of the latent GEMMs
 * (default), 1 a finishing kernel does; "convt_bwd" = 0 the 64 -> 64 ConvTranspose backward as one kernel that also makes the head's gradient
 * on the way (default), 1 as separate input- and weight-gradient kernels, 2 one kernel reading the stored gradient; "wgrad_batch" = 0
 * vae_model_backward computes the conv3x3 weight gradients of a backward stage in ONE launch (vae_conv3_wgrad_patch_multi: a seventh
 * of the split-K slab traffic; default), 1 one launch per layer; "conv1_twice" = 0 the whole-model passes
 * convolve the first layer twice and never store its output (vae_conv3x3_c8_stats / _bn_act / vae_bn_bwd_conv1_bits; default, needs
 * "conv1_bwd" = 0), 1 they store it; "reduce_direct" = 1 split-K slab folds of at most 8 slabs run one thread per 16-byte column with no
 * barrier (default), 0 they take the split-group form of the longer folds; "wgrad_ablate" = 0..7, only in a library built with
 * -DVAE_ABLATE (a no-op otherwise): vae_conv3_wgrad_patch on 32-wide images runs without its matrix instructions (bit 0),
 * its copies (bit 1), its fragment reads (bit 2) -- wrong results, for timing what bounds the kernel.  Unknown keys and
 * values outside a key's range are rejected.  Not thread-safe: set it while nothing is being launched, and never
 * between a vae_model_forward and its vae_model_backward: some keys change the workspace carving that
 * vae_model_workspace_bytes and both passes derive from the config. */
int vae_set_tuning(const char* key, int value);
/* Diagnostic aid: device buffer (>= 128 bytes per workgroup of the launch under study) that the diagnostic kernel
 * variants selected with vae_set_tuning write cycle-counter stamps to; NULL (default) = none. */
int vae_set_debug_buffer(void* dev_ptr);

/* ---- ELBO loss: vae_loss(), vae/training/train.py:44-51 -------------------------------- */
/* number of workgroups (= length of the `partial` array) the loss kernel uses for n_elems */
int vae_recon_loss_grid(long n_elems);
/* recon-term forward + backward in one pass: partial[b] = per-workgroup sum of the `kind` loss,
 * grad[i] = grad_scale * d loss_i / d recon_i  (grad may be NULL: forward only). */
int vae_recon_loss_fwd_bwd(const float* recon, const float* target, float* grad, float* partial,
                           long n_elems, int kind, float grad_scale, void* stream);
/* out3[0] = (sum(recon_partial) + kl_weight * sum(kl_partial)) * inv_batch   (train.py:50)
 * out3[1] = recon sum, out3[2] = KL sum */
int vae_elbo_combine(const float* recon_partial, int n_recon, const float* kl_partial, int n_kl,
                     float kl_weight, float inv_batch, float* out3, void* stream);

/* Allocation-free variant of vae_recon_loss_fwd_bwd + the fold of its partial sums.  No counterpart in the
 * reference; it exists for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): the per-workgroup partial sums live in a fixed 64 KB array inside the
 * library's code object (one copy per GPU, held until the process ends) instead of a caller buffer.
 *   out3[0] = out3[1] = sum of the `kind` loss over the n_elems elements (fixed summation order: the value
 *   vae_elbo_combine gives for the partials of vae_recon_loss_fwd_bwd), out3[2] = 0;  grad as above (may be NULL).
 * n_elems above vae_recon_loss_static_max_elems() (2^25 = 33 554 432: one partial sum per 2048 elements) is rejected.  It calls no hipMalloc / hipFree; work is
 * enqueued on `stream` only.  Every call on one GPU shares that GPU's array: calls that overlap on one GPU
 * (different streams, or host threads) race -- serialise them or use vae_recon_loss_fwd_bwd.  The first call on a
 * device resolves that device's copy of the array (hipGetSymbolAddress, which may load the code object): make it
 * outside any graph capture; every later call on that device only enqueues two kernels. */
long vae_recon_loss_static_max_elems(void);
int vae_recon_loss_fwd_bwd_static(const float* recon, const float* target, float* grad, float* out3,
                                  long n_elems, int kind, float grad_scale, void* stream);

/* ---- reparameterisation + KL: VAEConv.reparameterize, vae/models/vae.py:108-111, and the KL
 *      term of vae_loss, vae/training/train.py:47 ------------------------------------------ */
int vae_reparam_grid(long n_elems);
/* z = mu + exp(0.5 logvar) * eps; eps = eps_in if given, else Philox(seed, step, EPS stream, rank).
 * kl_partial[b] = per-workgroup share of -0.5*sum(1 + logvar - mu^2 - exp(logvar)).
 * z_f32 / z_bf16 / eps_out are optional outputs. */
int vae_reparam_kl_fwd(const float* mu, const float* logvar, const float* eps_in, float* z_f32,
                       void* z_bf16, float* eps_out, float* kl_partial, long n_elems,
                       unsigned long long seed, unsigned step, unsigned rank, void* stream);
/* The same fed by the split-K slabs of the fc_mu | fc_logvar GEMM (vae_linear_gemm_splitk with out == NULL):
 * mu | logvar [rows][latent] are OUTPUTS here, = bias[2 latent] + the nsplit slabs [s][row][2 latent] added in slab
 * order (exactly what the GEMM's own finishing step writes), then as above. */
int vae_reparam_kl_fwd_slabs(const float* slab, int nsplit, const float* bias, float* mu, float* logvar,
                             const float* eps_in, float* z_f32, void* z_bf16, float* eps_out, float* kl_partial,
                             long n_elems, int latent, unsigned long long seed, unsigned step, unsigned rank,
                             void* stream);
/* dmu = dz + dmu_ext + kl_scale*mu ; dlogvar = dz*eps*0.5*exp(0.5 logvar) + dlv_ext
 *       + kl_scale*0.5*(exp(logvar)-1).  dz, eps_in, dmu_ext, dlv_ext may be NULL.
 * dcat_bf16 (optional) = [row][dmu(0:L) | dlogvar(L:2L)]. */
int vae_reparam_kl_bwd(const float* dz, const float* mu, const float* logvar, const float* eps_in,
                       const float* dmu_ext, const float* dlv_ext, float kl_scale, float* dmu,
                       float* dlv, void* dcat_bf16, long n_elems, int latent,
                       unsigned long long seed, unsigned step, unsigned rank, void* stream);
/* The same with dz given as the nsplit split-K slabs [s][n_elems] of the GEMM that produces it (fc_up's input
 * gradient, vae_linear_gemm_splitk with out == NULL), added here in slab order; nsplit == 0: dz itself. */
int vae_reparam_kl_bwd_slabs(const float* dz, int nsplit, const float* mu, const float* logvar, const float* eps_in,
                             const float* dmu_ext, const float* dlv_ext, float kl_scale, float* dmu, float* dlv,
                             void* dcat_bf16, long n_elems, int latent, unsigned long long seed, unsigned step,
                             unsigned rank, void* stream);
/* raw streams, for RNG parity checks.  mode 0: uint32 words, 1: fp32 normals, 2: uint8 keep-mask
 * with drop probability p_drop (16-bit draws, 8 elements per Philox block; kept iff draw >=
 * vae_dropout_threshold(p) = ceil(p * 2^16)); 3: no stream but the normal generator's logarithm over its whole
 * domain, out[i] = ln((i + 1) 2^-24) for i < n <= 2^24 (the uniform of 24-bit draw i), so that it can be compared with a
 * host restatement value by value.  (torch.randn_like vae.py:110 / nn.Dropout vae.py:137) */
int vae_philox_fill(void* out, long n, int mode, double p_drop, unsigned long long seed,
                    unsigned step, unsigned stream_id, unsigned rank, void* stream);
unsigned vae_dropout_threshold(double p_drop);

/* ---- implicit-GEMM layers (bf16 MFMA, fp32 accumulate) ---------------------------------- */
/* Per-workgroup statistics: a GEMM over M rows and N columns leaves vae_gemm_stat_rows(M, N)
 * partial rows of N floats in stat_sum / stat_sq (one per workgroup row; the tile height is
 * 128 or, for small problems, 64 = vae_gemm_rows_per_block(), the bound to size buffers with). */
int vae_gemm_rows_per_block(void);
int vae_gemm_stat_rows(long M, int N);
/* partial-statistics rows vae_conv3x3_gemm writes for a geometry (its patch-staged kernel keeps
 * one row per workgroup, the generic kernel one per row tile); never more than
 * vae_gemm_stat_rows(n*H*W, Cout).  Belongs to the BatchNorm that follows the conv,
 * vae/models/vae.py:130,134. */
int vae_conv3x3_stat_rows(int n_img, int H, int W, int Cin, int Cout);
/* nn.Conv2d(Cin, Cout, 3, padding=1) forward, vae/models/vae.py:129,133 -- and, with the
 * flipped/transposed weight pack, its dgrad.  x [n,H,W,Cin], w_packed [Cout][9][Cin],
 * y [n,H,W,Cout].  stat_* (optional): vae_conv3x3_stat_rows partial rows of per-channel
 * sum / sum of squares of y. */
int vae_conv3x3_gemm(const void* x, const void* w_packed, const float* bias, void* y,
                     float* stat_sum, float* stat_sq, int n_img, int H, int W, int Cin, int Cout,
                     void* stream);
/* The first conv, nn.Conv2d(in_channels<=8, 64, 3, padding=1), vae/models/vae.py:34,129:
 * x8 is the network input as NHWC bf16 padded to 8 channels, w_packed [Cout][16 taps][8]
 * (taps 9..15 and channels >= in_channels are zero). */
int vae_conv3x3_c8_gemm(const void* x8, const void* w_packed, const float* bias, void* y,
                        float* stat_sum, float* stat_sq, int n_img, int H, int W, int Cout,
                        void* stream);
/* partial-statistics rows it writes (of Cout floats each); never more than vae_gemm_stat_rows(n*H*W, Cout) */
int vae_conv3x3_c8_stat_rows(int n_img, int H, int W, int Cout);
/* The same first conv followed by its BatchNorm2d + ReLU (vae/models/vae.py:129-131, the first three modules of the
 * first DoubleConv) as TWO passes over x8 that never store the conv output: the input is 16 bytes per pixel and the
 * output 2 * Cout, so convolving twice is cheaper than writing the conv output and reading it back.
 *   vae_conv3x3_c8_stats  -- the statistics rows of vae_conv3x3_c8_gemm (same values, same row count), nothing else;
 *   vae_conv3x3_c8_bn_act -- out [n,H,W,Cout] = relu(bf16(conv) * scale + shift), bit for bit what vae_bn_act_fwd makes of
 *                            the conv output vae_conv3x3_c8_gemm stores; relu_bits (optional): [n*H*W][Cout / 8] bytes, bit
 *                            (c & 7) of byte (c >> 3) of a pixel set where channel c passed the ReLU (for vae_bn_bwd_conv1_bits).
 * vae_conv3x3_c8_twice_ok: 1 when the geometry is covered (power-of-two images, Cout a multiple of 64, pixels a multiple of 32). */
int vae_conv3x3_c8_twice_ok(int n_img, int H, int W, int Cout);
int vae_conv3x3_c8_stats(const void* x8, const void* w_packed, const float* bias, float* stat_sum, float* stat_sq,
                         int n_img, int H, int W, int Cout, void* stream);
int vae_conv3x3_c8_bn_act(const void* x8, const void* w_packed, const float* bias, const float* scale, const float* shift,
                          void* out, void* relu_bits, int n_img, int H, int W, int Cout, void* stream);
/* nn.ConvTranspose2d(Cin, Cout, 2, stride=2) forward, vae/models/vae.py:64,67,70.
 * x [n,H,W,Cin], w_packed [(kh,kw,co)][Cin], y [n,2H,2W,Cout]; stats are per GEMM column
 * (tap*Cout + co), 4*Cout of them. */
int vae_convt2x2_fwd_gemm(const void* x, const void* w_packed, const float* bias, void* y,
                          float* stat_sum, float* stat_sq, int n_img, int H, int W, int Cin,
                          int Cout, void* stream);
/* partial-statistics rows it writes, of vae_convt2x2_stat_fold() * Cout floats each: `fold` column groups of Cout that the
 * BatchNorm behind it adds up (vae_bn_finalize's `fold`) -- 4, one per tap, unless a workgroup has summed taps itself
 * (then 2 or 1).  Rows: never more than vae_gemm_stat_rows(n*H*W, 4*Cout). */
int vae_convt2x2_stat_rows(int n_img, int H, int W, int Cin, int Cout);
int vae_convt2x2_stat_fold(int n_img, int H, int W, int Cin, int Cout);
/* its dgrad: dy [n,2H,2W,Cout], w_packed [Cin][(kh,kw,co)], dx [n,H,W,Cin] */
int vae_convt2x2_dgrad_gemm(const void* dy, const void* w_packed, void* dx, int n_img, int H,
                            int W, int Cin, int Cout, void* stream);
/* Input gradient AND weight gradient of a 64 -> 64 ConvTranspose2d(k2, s2) (vae/models/vae.py:131 with the default widths) in
 * one kernel that reads dy once: dx as vae_convt2x2_dgrad_gemm, the weight gradient as vae_wgrad_gemm(VAE_WGRAD_CONVT) --
 * vae_convt2x2_bwd_fused_slabs(n, H, W) split-K slabs [slab][co][tap][ci] of 4*64*64 floats, to be folded the same way.
 * dy [n,2H,2W,64], x [n,H,W,64] bf16; w_packed: the operand of vae_convt2x2_dgrad_gemm.  n*H*W must be a multiple of 64;
 * vae_convt2x2_bwd_fused_ok says whether the geometry is covered (and the "convt_bwd" tuning key leaves it on). */
int vae_convt2x2_bwd_fused_ok(int n_img, int H, int W, int Cin, int Cout);
int vae_convt2x2_bwd_fused_slabs(int n_img, int H, int W);
int vae_convt2x2_bwd_fused(const void* dy, const void* x, const void* w_packed, void* dx, float* slab, int n_img,
                           int H, int W, int Cin, int Cout, void* stream);
/* nn.Linear forward / dgrad (fc_mu | fc_logvar fused, fc_up), vae/models/vae.py:48-49,63,96-97,116:
 * out[m][n] = sum_k a[m][k] w[n][k] (+ bias[n]).  K % 64 == 0, or K % 32 == 0 with VAE_EPI_BF16. */
int vae_linear_gemm(const void* a, const void* w, const float* bias, void* out, void* out2,
                    float* stat_sum, float* stat_sq, int M, int N, int K, int epilogue,
                    void* stream);
/* The same nn.Linear forward / dgrad (vae/models/vae.py:96-97,116) for the fp32 epilogues when M
 * is only the batch: the K range is cut into nsplit slabs [nsplit][M][N] (fp32, caller-provided)
 * computed by separate workgroups and summed, with the bias, in a fixed order.  out == NULL: the slabs are left
 * for the caller's next kernel to add up (vae_reparam_kl_fwd_slabs / vae_reparam_kl_bwd_slabs).
 * vae_linear_splitk suggests nsplit (1 = not worth splitting). */
int vae_linear_splitk(int M, int N, int K);
int vae_linear_gemm_splitk(const void* a, const void* w, const float* bias, float* out, float* out2,
                           float* slab, int nsplit, int M, int N, int K, int epilogue,
                           void* stream);

/* ---- weight gradients (what autograd computes for the layers above in loss.backward(),
 *      vae/training/train.py:332) ------------------------------------------------------------ */
#define VAE_WGRAD_PLAIN 0     /* dW[m][n]      = sum_k P[k][m] Q[k][n]                 (nn.Linear)          */
#define VAE_WGRAD_CONV3 1     /* dW[m][tap][n] = sum_pix dY[pix][m] X[pix+tap][n]      (Conv2d 3x3 p1)      */
#define VAE_WGRAD_CONVT 2     /* dW[m][tap][n] = sum_pix dY[2i+kh,2j+kw][m] X[pix][n]  (ConvTranspose k2s2) */
#define VAE_WGRAD_CONV3_C8 3  /* first conv: Q = 8-channel input, dW[m][16 taps][8]                          */
/* suggested K split for a problem (keeps the chip full) */
int vae_wgrad_ksplit(int M, int N, int taps, long Ktot);
/* P [K rows][ldp] (M channels used), Q [K rows][ldq] (N channels used), both bf16.
 * slab: nsplit x taps x M x N floats, written as [split][m][tap][n].  H, W: pixel grid of K
 * (ignored for PLAIN). */
int vae_wgrad_gemm(const void* P, const void* Q, float* slab, int M, int N, long Ktot, int ldp,
                   int ldq, int mode, int H, int W, int nsplit, void* stream);
/* Conv2d 3x3 weight gradient with all nine taps per workgroup (one staged input patch serves
 * the nine shifts).  Same slab layout as VAE_WGRAD_CONV3: [split][Cout][9][Cin].  Covers
 * power-of-two images with 4 <= W <= 64 and channel counts that are multiples of 64
 * (vae_conv3_wgrad_patch_ok); other shapes go through vae_wgrad_gemm. */
int vae_conv3_wgrad_patch_ok(int H, int W, int Cin, int Cout);
int vae_conv3_wgrad_patch_ksplit(int n_img, int H, int W, int Cin, int Cout);
int vae_conv3_wgrad_patch(const void* dy, const void* x, float* slab, int n_img, int H, int W,
                          int Cin, int Cout, int nsplit, void* stream);
/* The same for several conv layers in ONE launch (the Conv2d weight gradients of a backward stage, vae/models/vae.py:129,133
 * x 2-4 layers): the launch's workgroups are dealt out over the jobs in proportion to their work.  A launch that fills
 * the chip writes one 147 KB slab per CU whatever it computes, so batching divides the split-K slab traffic by the
 * number of jobs.  vae_conv3_wgrad_multi_plan fills nsplit_out[i] so that the launch has at most max_wgs workgroups
 * (<= 0: 256, one per CU) with about equal K ranges; job i needs nsplit[i] * 9 * Cout * Cin floats of slab, laid out as
 * vae_conv3_wgrad_patch lays it out.  Every job's geometry must pass vae_conv3_wgrad_patch_ok. */
#define VAE_MAX_WGRAD_JOBS 8
typedef struct vae_wgrad_job {
    const void* dy;     /* [n_img*H*W][Cout] bf16 */
    const void* x;      /* [n_img*H*W][Cin]  bf16 */
    float* slab;
    int n_img, H, W, Cin, Cout, nsplit;
} vae_wgrad_job;
int vae_conv3_wgrad_multi_plan(const vae_wgrad_job* jobs, int njobs, int max_wgs, int* nsplit_out);
int vae_conv3_wgrad_patch_multi(const vae_wgrad_job* jobs, int njobs, void* stream);
/* out[e] (+)= sum_s slab[s*slab_stride + base + i0*s0+i1*s1+i2*s2+i3*s3] where e is the row-major
 * index over dims4 = (D0,D1,D2,D3): folds the K splits and converts the kernels' packed
 * layout to the parameter's own layout (e.g. Conv2d [Cout][Cin][3][3]). */
int vae_reduce_slabs(const float* slab, int nsplit, long slab_stride, float* out, const long* dims4,
                     const long* strides4, long base, int accumulate, void* stream);
/* The same fold walking the slab in ITS order (coalesced) and scattering the results:
 * out[base + i0*s0+..+i3*s3] (+)= sum_s slab[s*slab_stride + slab_base + q], q row-major over
 * dims4 (the slab's own dims); tuples with i_d >= limits4[d] (padding) are skipped. */
int vae_reduce_slabs_scatter(const float* slab, int nsplit, long slab_stride, long slab_base,
                             float* out, const long* dims4, const long* strides4,
                             const long* limits4, long base, int accumulate, void* stream);
/* Several vae_reduce_slabs_scatter folds in one launch (all the weight gradients of a backward stage:
 * the folds are a few microseconds each and only the optimizer waits for them). */
#define VAE_MAX_REDUCE_JOBS 20
typedef struct vae_reduce_job {
    const float* slab;
    int nsplit;
    long slab_stride, slab_base;
    float* out;
    long dims[4], strides[4], limits[4];   /* as in vae_reduce_slabs_scatter; limits = dims when nothing is padding */
    long base;
    int accumulate;
} vae_reduce_job;
int vae_multi_reduce_slabs_scatter(const vae_reduce_job* jobs, int njobs, void* stream);
/* Many pack / permute jobs in one launch (all weights of the model after an optimizer step). */
#define VAE_MAX_PACK_JOBS 40
typedef struct vae_pack_job {
    const float* src;
    void* dst;          /* bf16 (f32 = 0) or float (f32 = 1) */
    long dims[4];       /* row-major index space of src */
    long strides[4];    /* dst offset = base + sum i_d * strides[d] */
    long base;
    int f32;
} vae_pack_job;
int vae_multi_pack(const vae_pack_job* jobs, int njobs, void* stream);
/* dst_bf16[base + i0*s0+i1*s1+i2*s2+i3*s3] = bf16(src[e]): fp32 parameter -> packed bf16 operand */
int vae_pack_bf16(const float* src, void* dst_bf16, const long* dims4, const long* strides4,
                  long base, void* stream);

/* ---- BatchNorm2d + ReLU + Dropout + MaxPool2d(2): the tail of DoubleConv and the pools,
 *      vae/models/vae.py:130-137,35,38,41, and bn3/bn2/bn1 + relu, vae.py:119-121 ------------- */
/* Folds the GEMM epilogue's per-workgroup sums (rows x fold*C) into scale = gamma*rstd,
 * shift = beta - mean*scale, mean, rstd, and moves running_mean / running_var (momentum 0.1,
 * unbiased variance) when they are given.  use_running = 1: eval mode, stats are not read.
 * stat_sum / stat_sq must have VAE_STAT_SCRATCH_ROWS more rows than `rows`: with many partial
 * rows they are first folded into that tail by the whole chip. */
#define VAE_STAT_SCRATCH_ROWS 32
int vae_bn_finalize(float* stat_sum, float* stat_sq, int rows, int C, int fold,
                    double count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, int use_running, float* scale,
                    float* shift, float* mean, float* rstd, void* stream);
/* out[j] = mul * sum_r partial[r][j] */
int vae_sum_partials(const float* partial, int rows, int ncols, float* out, float mul, void* stream);
/* y [n,H,W,C] -> out [n,H,W,C] or, with pool, [n,H/2,W/2,C].  Dropout keep flags come from
 * keep_mask (uint8, y's layout) when given, else from Philox(seed, step, stream_id, rank) at the
 * element's NHWC linear index. */
int vae_bn_act_fwd(const void* y, const float* scale, const float* shift, void* out, int n_img,
                   int H, int W, int C, int pool, double p_drop, const void* keep_mask,
                   unsigned long long seed, unsigned step, unsigned stream_id, unsigned rank,
                   void* stream);
/* The same with a route record for the pooled form (nn.MaxPool2d(2) after nn.Dropout, vae.py:35-41,137):
 * route [n,H/2,W/2,C] uint8, one byte per pooled element -- bits 0-1 the window position (scan order,
 * first maximum wins) of the maximum, bit 2 set when the maximum is positive, i.e. when a gradient flows
 * through it.  route may be null (nothing is recorded); it must be null when pool == 0. */
int vae_bn_act_fwd_route(const void* y, const float* scale, const float* shift, void* out, void* route,
                         int n_img, int H, int W, int C, int pool, double p_drop, const void* keep_mask,
                         unsigned long long seed, unsigned step, unsigned stream_id, unsigned rank,
                         void* stream);
/* The same, also leaving ymax [n,H/2,W/2,C] bf16: the conv output y at each window's maximum (null: not written; it needs
 * route).  vae_bn_act_bwd_route_y then reduces its per-channel sums from ymax instead of re-reading the four window
 * elements of y: a quarter of that pass's reads of y, bit-identical sums (the values are the same bf16 numbers). */
int vae_bn_act_fwd_route_y(const void* y, const float* scale, const float* shift, void* out, void* route, void* ymax,
                           int n_img, int H, int W, int C, int pool, double p_drop, const void* keep_mask,
                           unsigned long long seed, unsigned step, unsigned stream_id, unsigned rank,
                           void* stream);
/* rows of the `partial` workspace (x 2*C floats) used by vae_bn_act_bwd */
int vae_bn_act_bwd_blocks(int n_img, int H, int W, int C, int pool);
/* backward of the above down to the conv output: gin = grad of the (pooled) block output,
 * dy = grad of the pre-norm tensor y; sums = [d gamma (C) | d beta (C)]. */
int vae_bn_act_bwd(const void* gin, const void* y, const float* scale, const float* shift,
                   const float* mean, const float* rstd, float* partial, float* sums, void* dy,
                   int n_img, int H, int W, int C, int pool, double p_drop, const void* keep_mask,
                   unsigned long long seed, unsigned step, unsigned stream_id, unsigned rank,
                   void* stream);
/* The same reading the route bytes vae_bn_act_fwd_route left (pooled form): no dropout stream is
 * regenerated and no activation recomputed; with route == null it is vae_bn_act_bwd. */
int vae_bn_act_bwd_route(const void* gin, const void* y, const float* scale, const float* shift,
                         const float* mean, const float* rstd, float* partial, float* sums, void* dy,
                         const void* route, int n_img, int H, int W, int C, int pool, double p_drop,
                         const void* keep_mask, unsigned long long seed, unsigned step,
                         unsigned stream_id, unsigned rank, void* stream);
/* vae_bn_act_bwd_route with the ymax tensor vae_bn_act_fwd_route_y left (null: as vae_bn_act_bwd_route). */
int vae_bn_act_bwd_route_y(const void* gin, const void* y, const float* scale, const float* shift,
                           const float* mean, const float* rstd, float* partial, float* sums, void* dy,
                           const void* route, const void* ymax, int n_img, int H, int W, int C, int pool,
                           double p_drop, const void* keep_mask, unsigned long long seed, unsigned step,
                           unsigned stream_id, unsigned rank, void* stream);
/* Backward of the network's first  Conv2d(Cin, 64, 3, padding=1) -> BatchNorm2d(64) -> ReLU  (vae/models/vae.py:24-31
 * with in_channels = the image channels, vae.py:34) in one pass over gin and y.  Nothing in front of that conv takes
 * an input gradient, so the gradient of the conv output is never stored: the kernel accumulates per channel
 * sum(d * xhat), sum(d) (d = gin where the ReLU passed) and, on the matrix pipe, sum_p d X, sum_p y X and sum_p X against
 * the nine shifted copies X of the input; BatchNorm's backward is affine in those, so
 *   dW = scale * (sum d X - mean(d) sum X - mean(d xhat) * rstd * (sum y X - mean * sum X)).
 * gin, y [n,H,W,64] bf16; x8 [n,H,W,8] bf16 (vae_nchw_to_nhwc8); scale/shift/mean/rstd [64] as left by vae_bn_finalize;
 * scratch: vae_bn_bwd_conv1_scratch_floats(n, H, W) floats, 16-byte aligned; sums = [d gamma (64) | d beta (64)];
 * dW [64][Cin][3][3] f32 (overwritten).  Cin <= 3 (nine taps x Cin columns must fit 32).  Same result as
 * vae_bn_act_bwd followed by vae_wgrad_gemm(VAE_WGRAD_CONV3_C8) up to rounding: there g is rounded to bf16 before
 * the sum over pixels, here the three sums are exact products accumulated in fp32 and folded in fp64. */
long vae_bn_bwd_conv1_scratch_floats(int n_img, int H, int W);
int vae_bn_bwd_conv1(const void* gin, const void* y, const void* x8, const float* scale, const float* shift,
                     const float* mean, const float* rstd, float* scratch, float* sums, float* dW, int n_img,
                     int H, int W, int Cin, void* stream);
/* The same backward (vae/models/vae.py:24-31,34) when the forward ran the first conv in two passes and never stored its
 * output: relu_bits [n*H*W][8] (vae_conv3x3_c8_bn_act) holds the ReLU decisions, and each 64-pixel tile's y is rebuilt
 * on the matrix pipe from the tile's shifted-input rows and the conv's own parameters w [64][Cin][3][3] / bias [64]
 * (fp32, rounded to bf16 as the forward's operand pack rounds them).  A rebuilt y can differ from the forward's in its
 * last bf16 bit (other order of the 27 products); it only enters sums over all pixels.  scale / mean / rstd, scratch,
 * sums and dW as above. */
int vae_bn_bwd_conv1_bits(const void* gin, const void* relu_bits, const void* x8, const float* w, const float* bias,
                          const float* scale, const float* mean, const float* rstd, float* scratch, float* sums,
                          float* dW, int n_img, int H, int W, int Cin, void* stream);
/* ---- GroupNorm form of the double-conv block: nn.GroupNorm(8, C) + ReLU (+ nn.Dropout) (+ nn.MaxPool2d(2)),
 *      vae/practice/vae_conv.py:20-36,50-59 and vae/practice/vae_original.py:58-74 ----------------------------
 * y [n,H,W,C] bf16 (a conv output) -> out [n,H,W,C] or, with pool, [n,H/2,W/2,C].  Statistics are per image and
 * group (biased variance, eps inside the square root), the same in train and eval mode.  mean_rstd
 * [n][groups][2] f32 receives (mean, rstd) for the backward pass (may be null for inference).  C must be a
 * multiple of 8*groups, at most 512.  Dropout flags as in vae_bn_act_fwd. */
int vae_gn_act_fwd(const void* y, const float* gamma, const float* beta, void* out, float* mean_rstd,
                   int n_img, int H, int W, int C, int groups, float eps, int pool, double p_drop,
                   const void* keep_mask, unsigned long long seed, unsigned step, unsigned stream_id,
                   unsigned rank, void* stream);
/* its backward down to the conv output: gin = grad of the (pooled) block output, dy = grad of y;
 * chan_partial [n][3][C] f32 scratch (per-image sums), sums = [d gamma (C) | d beta (C) | sum of dy per
 * channel (C), i.e. the bias gradient of the conv that produced y]. */
int vae_gn_act_bwd(const void* gin, const void* y, const float* gamma, const float* beta,
                   const float* mean_rstd, float* chan_partial, float* sums, void* dy, int n_img,
                   int H, int W, int C, int groups, int pool, double p_drop, const void* keep_mask,
                   unsigned long long seed, unsigned step, unsigned stream_id, unsigned rank, void* stream);
/* ---- nn.MaxPool2d(2) on its own, vae/models/vae.py:35,38,41 (the model's pooling is fused into the
 *      activation kernels above; the drafts' U-Net keeps the unpooled tensor too,
 *      vae/practice/vae_original.py:142-149) --------------------------------------------------------------
 * x [n,H,W,C] bf16 -> out [n,H/2,W/2,C]; route (optional, [n,H/2,W/2,C] uint8) receives the window position
 * (scan order, first maximum wins) of each maximum. */
int vae_maxpool2_fwd(const void* x, void* out, void* route, int n_img, int H, int W, int C, void* stream);
/* dx [n,H,W,C] = gout at each window's maximum, 0 elsewhere; the maximum comes from route when given
 * (x may then be null), else it is recomputed from x. */
int vae_maxpool2_bwd(const void* gout, const void* x, const void* route, void* dx, int n_img, int H, int W,
                     int C, void* stream);
/* out[c] = sum_r x[r][c] of a [rows][C] bf16 matrix: the bias gradient of a conv / transposed conv from the
 * gradient of its output (nn.ConvTranspose2d bias in the drafts' U-Net decoders, vae/practice/vae_original.py:110-116,
 * where no normalisation follows and the gradient is not zero).  partial: vae_colsum_blocks(rows, C) x C floats. */
int vae_colsum_blocks(long rows, int C);
int vae_colsum_bf16(const void* x, long rows, int C, float* partial, float* out, void* stream);
/* relu(bn1(.)) -> nn.Conv2d(64, Cout, 1) -> sigmoid, vae/models/vae.py:74,121-122.
 * y [n,H,W,64] bf16, w [Cout][64] f32, recon [n,Cout,H,W] f32 (NCHW, the module output). */
int vae_head_fwd(const void* y, const float* scale, const float* shift, const float* w,
                 const float* b, float* recon, int n_img, int H, int W, int Cout, void* stream);
int vae_head_bwd_blocks(int n_img, int H, int W);
/* its backward: partial is blocks x (3+Cout)*64 floats of scratch, 16-byte aligned; sums = [d gamma1(64) |
 * d beta1(64)]; sums_head = [dW[Cout][64] | db(Cout)] (Cout*65 floats);
 * dy [n,H,W,64] bf16 = grad of up1's output. */
int vae_head_bwd(const float* drecon, const float* recon, const void* y, const float* scale,
                 const float* shift, const float* mean, const float* rstd, const float* w,
                 float* partial, float* sums, float* sums_head, void* dy, int n_img, int H, int W,
                 int Cout, void* stream);
/* The sums of vae_head_bwd alone (sums, sums_head as there; no dy), also leaving the logit gradients
 * dl [n*H*W][4] f32 = drecon * recon * (1 - recon) (entries >= Cout are 0): the input of vae_head_convt_bwd_fused. */
int vae_head_bwd_sums(const float* drecon, const float* recon, const void* y, const float* scale,
                      const float* shift, const float* mean, const float* rstd, const float* w,
                      float* partial, float* sums, float* sums_head, float* dl, int n_img, int H, int W,
                      int Cout, void* stream);
/* vae_head_bwd's second pass and vae_convt2x2_bwd_fused in one kernel (vae/models/vae.py:131-134, the default 64 -> 64
 * widths): the gradient dy behind bn1 + ReLU + head is computed from y [n,2H,2W,64] (the ConvTranspose output), dl and
 * the sums of vae_head_bwd_sums, tile by tile in LDS, and never stored; dx [n,H,W,64] and the weight-gradient slabs
 * come out as from vae_convt2x2_bwd_fused (same slab count).  x [n,H,W,64]: the ConvTranspose input. */
int vae_head_convt_bwd_fused(const void* y, const float* dl, const float* scale, const float* shift, const float* mean,
                             const float* rstd, const float* sums, const float* w_head, int Cout_head, const void* x,
                             const void* w_packed, void* dx, float* slab, int n_img, int H, int W, void* stream);
/* torch.clamp(x, lo, hi) on fp32 (logvar in the deeper U-Net draft, vae/practice/vae_original.py:390) and its
 * backward: g null -> out = clamp(x); g given -> out = g where lo <= x <= hi, else 0. */
int vae_clamp_f32(const float* x, const float* g, float lo, float hi, float* out, long n, void* stream);
/* nn.Conv2d(64, Cout, 1) alone, logits out: the output layer of the drafts' U-Net models
 * (vae/practice/vae_original.py:119,173-175; no normalisation in front, no sigmoid behind).
 * x [n,H,W,64] bf16, w [Cout][64] f32, b [Cout], out [n,Cout,H,W] f32 (NCHW).  Cout in {1, 3, 4}. */
int vae_conv1x1_out_fwd(const void* x, const float* w, const float* b, float* out, int n_img, int H, int W,
                        int Cout, void* stream);
int vae_conv1x1_out_bwd_blocks(int n_img, int H, int W);
/* its backward: gout [n,Cout,H,W] f32; partial is blocks x (Cout+1)*64 floats of scratch;
 * sums = [dW[Cout][64] | db(Cout) padded to 64]; dx [n,H,W,64] bf16 (may be null). */
int vae_conv1x1_out_bwd(const float* gout, const void* x, const float* w, float* partial, float* sums,
                        void* dx, int n_img, int H, int W, int Cout, void* stream);
/* torch.tanh backward (vae.py:116): dpre = dx*(1-h^2), colpartial [vae_tanh_bwd_rows(rows)][cols]
 * = per-row-group column sums of dpre (fc_up bias gradient). */
int vae_tanh_bwd_rows(int rows);
int vae_tanh_bwd(const void* dx, const void* h, void* dpre, float* colpartial, int rows, int cols,
                 void* stream);
/* inputs.to(device) (train.py:325): NCHW f32 image batch -> NHWC bf16 padded to 8 channels */
int vae_nchw_to_nhwc8(const float* x, void* out, int n_img, int C, int H, int W, void* stream);
/* the same, also adding 1 to counters[0 .. ncount) (int64, device; BatchNorm2d.num_batches_tracked of a train-mode
 * forward pass): the first kernel of the pass does it on the way.  counters may be null. */
int vae_nchw_to_nhwc8_count(const float* x, void* out, int n_img, int C, int H, int W, long long* counters, int ncount,
                            void* stream);
/* Batch feed from a dataset resident in HBM: out[b][c][h][w] = data[index[b]][h][w][c] / 255 (fp32, the
 * value torchvision's ToTensor gives).  Replaces the host-side batch assembly of the reference's DataLoader
 * (vae/training/train.py:158-179: random_split + DataLoader(shuffle); :266-274: CIFAR10 / ToTensor).
 * data: n_data images, uint8, HWC; index: batch int64 image indices (device memory; negative values count
 * from the end; null = images 0..batch-1); an index outside the dataset yields a zero image. */
int vae_feed_u8(const void* data_u8_nhwc, long n_data, const long* index, float* out_nchw, int batch, int C,
                int H, int W, void* stream);
/* dst[base + i0*s0+i1*s1+i2*s2+i3*s3] = src[e] (fp32 permuting copy); dst_bf16[i] = bf16(src[i]) */
int vae_permute_f32(const float* src, float* dst, const long* dims4, const long* strides4, long base,
                    void* stream);
int vae_cast_bf16(const float* src, void* dst_bf16, long n, void* stream);

/* ---- whole model: VAEConv(in_channels, out_channels, latent_dims, p_dropout, image_size),
 *      vae/models/vae.py:22-122, as one call per direction --------------------------------------
 * params / grads : ONE flat fp32 buffer each, the 52 parameters in state_dict order with the
 *                  reference shapes, tensor i at vae_model_param_offset(cfg, i) (16-byte aligned).
 * bn_stats       : flat fp32 [running_mean(C) | running_var(C)] for the 11 BatchNorms in module
 *                  order (conv1.net.1, conv1.net.4, ... conv4.net.4, bn3, bn2, bn1).
 * wpack          : bf16 operand copies of the weights, vae_model_wpack_bytes(cfg) bytes, refreshed
 *                  by vae_model_pack_weights after every parameter update.  Zero-fill it once when
 *                  it is allocated: padding lanes of the first conv's operand are never written.
 * workspace      : vae_model_workspace_bytes(cfg) bytes; holds the activations between forward
 *                  and backward, so one workspace per in-flight batch. */
typedef struct vae_model_cfg {
    unsigned struct_size;    /* = sizeof(vae_model_cfg) of the header the caller was built against; every vae_model_*
                              * entry point rejects a config whose size differs from the library's (a caller built
                              * against another revision of this struct would otherwise hand over garbage fields) */
    int batch, in_channels, out_channels, latent, image_size;
    int train;               /* 1: model.train() -- batch statistics, dropout, running stats move */
    double p_dropout;
    unsigned long long seed; /* Philox key (config seed) */
    unsigned step, rank;     /* Philox counter words: forward-call index and data-parallel rank */
    void* side_stream;       /* optional second hipStream_t: vae_model_backward runs the weight-gradient
                              * kernels there, beside the dgrad / BatchNorm chain on `stream`, and
                              * re-joins `stream` before it returns.  NULL: everything on `stream`. */
    int precision;           /* 0 (VAE_PRECISION_BF16): bf16 MFMA operands and bf16 activations, fp32 accumulation -- the
                              * throughput path.  1 (VAE_PRECISION_FP32): the reference's own arithmetic type
                              * (vae/training/train.py:430-431 trains in fp32): NCHW fp32 tensors, parameters read in
                              * their own layouts (wpack is not used and may be NULL), sums of products accumulated in
                              * fp64 and rounded to fp32 once.  Same entry points, same params / grads / bn_stats
                              * buffers; vae_model_workspace_bytes depends on it.  In this mode vae_model_backward
                              * writes all 52 gradients (the biases in front of a BatchNorm get their rounding-noise
                              * sums, as in the reference) and ignores side_stream. */
    long long* batches_tracked;  /* optional, device: the eleven BatchNorm2d.num_batches_tracked counters in module order
                              * (conv1.net.1 ... conv4.net.4, bn3, bn2, bn1).  With train != 0 and the bf16 path,
                              * vae_model_forward adds 1 to all of them and vae_model_encode to the first eight, as
                              * nn.BatchNorm2d.forward does in train mode (vae/models/vae.py:26,32,57-72), inside a
                              * kernel it launches anyway.  NULL (or the fp32 path): the caller keeps the counters. */
} vae_model_cfg;
#define VAE_PRECISION_BF16 0
#define VAE_PRECISION_FP32 1

long vae_model_param_count(const vae_model_cfg* cfg);              /* floats in params / grads */
long vae_model_param_offset(const vae_model_cfg* cfg, int index);  /* index 0..51 */
long vae_model_bn_count(const vae_model_cfg* cfg);
long vae_model_wpack_bytes(const vae_model_cfg* cfg);
long vae_model_workspace_bytes(const vae_model_cfg* cfg);
int vae_model_pack_weights(const vae_model_cfg* cfg, const float* params, void* wpack, void* stream);
/* VAEConv.forward, vae.py:101-105: x [N,C,S,S] f32 -> recon [N,Cout,S,S], mu, logvar [N,L] f32.
 * eps_in (optional) replaces the Philox draw of torch.randn_like (vae.py:110); keep_masks
 * (optional) = 4 uint8 NHWC keep-masks replacing the Philox draws of nn.Dropout (vae.py:137). */
int vae_model_forward(const vae_model_cfg* cfg, const float* params, float* bn_stats, const void* wpack,
                      void* workspace, const float* x, const float* eps_in,
                      const void* const* keep_masks, float* recon, float* mu, float* logvar,
                      void* stream);
/* VAEConv.encode, vae.py:89-98 and VAEConv.decode, vae.py:114-122 */
int vae_model_encode(const vae_model_cfg* cfg, const float* params, float* bn_stats, const void* wpack,
                     void* workspace, const float* x, const void* const* keep_masks, float* mu,
                     float* logvar, void* stream);
int vae_model_decode(const vae_model_cfg* cfg, const float* params, float* bn_stats, const void* wpack,
                     void* workspace, const float* z, float* recon, void* stream);
/* the per-workgroup KL partial sums the last vae_model_forward left in the workspace */
int vae_model_kl_partials(const vae_model_cfg* cfg, void* workspace, const float** kl_partial, int* n);
/* loss.backward(), train.py:332, for the whole module: grads of all 52 parameters from
 * d_recon (= d loss / d recon), optional external d loss/d mu, d loss/d logvar, and
 * kl_scale * KL (kl_scale = kl_weight / batch when the KL term is not in dmu_ext/dlv_ext).
 * stage -1 = everything; 0 = head..fc_mu/fc_logvar, 1 = conv4+conv3, 2 = conv2+conv1 (in that
 * order) so that gradient all-reduce of finished layers can overlap the rest.
 * grads must be ZERO when the first stage of a pass starts: the biases in front of a train-mode
 * BatchNorm (conv*.net.{0,3}.bias, up*.bias) have an identically zero gradient and their slots
 * are not written.  Every other slot is overwritten, not accumulated into. */
int vae_model_backward(const vae_model_cfg* cfg, const float* params, const void* wpack, void* workspace,
                       const float* d_recon, const float* recon, const float* mu, const float* logvar,
                       const float* eps_in, const void* const* keep_masks, const float* dmu_ext,
                       const float* dlv_ext, float kl_scale, float* grads, int stage, void* stream);

/* ---- clip_grad_norm_(max_norm) + AdamW.step(), vae/training/train.py:336,338,310, on the flat
 *      buffers.  grads are first multiplied by grad_mul (1/world_size after a SUM all-reduce);
 *      max_norm <= 0 disables clipping; partial: vae_adamw_grid(n) floats of scratch;
 *      grad_norm_out (optional) receives the pre-clip global L2 norm. */
int vae_adamw_grid(long n);
int vae_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                        float* partial, float grad_mul, float max_norm, double lr, double beta1,
                        double beta2, double eps, double weight_decay, int step, float* grad_norm_out,
                        void* stream);
/* The same, with vae_elbo_combine (train.py:44-51, the step's loss value) taken along in the first launch when out3 is
 * not NULL: the loss sums have no reader on the device, so the step saves their launch. */
int vae_adamw_clip_step_elbo(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n,
                             float* partial, float grad_mul, float max_norm, double lr, double beta1,
                             double beta2, double eps, double weight_decay, int step, float* grad_norm_out,
                             const float* recon_partial, int n_recon, const float* kl_partial, int n_kl,
                             float kl_weight, float inv_batch, float* out3, void* stream);

#ifdef __cplusplus
}
#endif
#endif
