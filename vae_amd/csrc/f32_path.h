// Kernels of the fp32 execution mode of VAEConv (vae_model_cfg.precision = 1): the reference computes in fp32
// (vae/training/train.py:430-431, vae/models/vae.py:101-122), and this mode reproduces it on the GPU to fp32 rounding:
// NCHW fp32 tensors and the parameters' own layouts (no packed copies), products of fp32 operands summed in fp64 and
// rounded to fp32 once, BatchNorm statistics in fp64.  It is the precision mode, not the fast one: the bf16 MFMA
// kernels are the throughput path.
#pragma once
#include <hip/hip_runtime.h>

#include "act.h"

namespace vae {
namespace f32 {

// C[m][n] = sum_k A(m, k) * B(k, n): one tiled kernel, the operand index maps chosen by mode
enum Mode {
    CONV_FWD,     // nn.Conv2d(Ci, Co, ks, padding=ks/2) forward      a = x [n][Ci][H][W], b = weight [Co][Ci][ks][ks] -> y [n][Co][H][W]
    CONV_DGRAD,   // its input gradient                               a = dy [n][Co][H][W], b = weight -> dx [n][Ci][H][W]
    CONV_WGRAD,   // its weight gradient                              a = dy, b = x -> dW [Co][Ci][ks][ks]
    CONVT_FWD,    // nn.ConvTranspose2d(Ci, Co, 2, stride=2) forward  a = x [n][Ci][H][W], b = weight [Ci][Co][2][2] -> y [n][Co][2H][2W]
    CONVT_DGRAD,  // its input gradient                               a = dy [n][Co][2H][2W], b = weight -> dx [n][Ci][H][W]
    CONVT_WGRAD,  // its weight gradient                              a = x, b = dy -> dW [Ci][Co][2][2]
    LIN_FWD,      // nn.Linear(in = Ci, out = Co) forward             a = x [n][Ci], b = weight [Co][Ci] -> y [n][Co]
    LIN_DGRAD,    // its input gradient                               a = dy [n][Co], b = weight -> dx [n][Ci]
    LIN_WGRAD     // its weight gradient                              a = dy [n][Co], b = x [n][Ci] -> dW [Co][Ci]
};
enum Epilogue { EPI_NONE = 0, EPI_TANH = 1, EPI_SIGMOID = 2 };

struct Gemm {
    const float* a;
    const float* b;
    const float* bias;     // forward modes: per output channel / feature, or null
    float* out;
    int n_img, H, W;       // batch and the conv's grid (ConvTranspose: its INPUT grid; Linear: H = W = 1)
    int Ci, Co, ks;        // ks: 3 or 1 (conv modes)
    int epi;               // forward modes
    int accumulate;        // dgrad modes: out += result
    float* slab;           // split-K scratch of slab_need() floats (8-byte aligned; null where slab_need() is 0)
};
// floats of split-K scratch this geometry needs (0 when its output tiles fill the chip without a split)
long slab_need(int mode, int n_img, int H, int W, int Ci, int Co, int ks);
int gemm(int mode, const Gemm& g, hipStream_t s);

// BatchNorm2d statistics of y [n][C][HW] -> coef: scale[C] | shift[C] | mean[C] | rstd[C] (each `coef_stride` floats
// apart); train: batch statistics, running stats move (momentum, unbiased variance); else running stats are used
int bn_stats(const float* y, int n_img, int C, int HW, const float* gamma, const float* beta, float* running_mean,
             float* running_var, int train, float momentum, float eps, float* coef, int coef_stride, double* part,
             hipStream_t s);
// out = [maxpool2] dropout(relu(y * scale + shift)); argmax (pooled only): window position 0..3 of the maximum
int bn_act_fwd(const float* y, const float* coef, int coef_stride, float* out, unsigned char* argmax, int n_img, int C,
               int H, int W, int pool, const DropCfg& drop, hipStream_t s);
// g = d loss / d out -> dy (= d loss / d y), d gamma, d beta; sums: 2 * C floats of scratch
int bn_act_bwd(const float* g, const float* y, const float* coef, int coef_stride, const float* gamma,
               const unsigned char* argmax, float* dy, float* dgamma, float* dbeta, float* sums, double* part, int n_img, int C,
               int H, int W, int pool, const DropCfg& drop, hipStream_t s);
// out[c] = sum over images and positions of t [n][C][HW] (bias gradients; HW = 1: column sums of a matrix)
int chan_sum(const float* t, int n_img, int C, int HW, float* out, double* part, hipStream_t s);
// `part`: scratch of chan_part_doubles(C) doubles for the two-stage per-channel reductions above (one workgroup per
// (channel, share of the images), then one thread per channel)
long chan_part_doubles(int C);
int tanh_bwd(const float* dh, const float* h, float* dpre, long n, hipStream_t s);        // dpre = dh * (1 - h^2)
int sigmoid_bwd(const float* dr, const float* r, float* dpre, long n, hipStream_t s);     // dpre = dr * r * (1 - r)

}  // namespace f32
}  // namespace vae
