// Device-side batch feed: the dataset lives in HBM as uint8 images in HWC order (the order PIL / the CIFAR
// arrays give), a batch is a gather by image index fused with torchvision's ToTensor: HWC uint8 -> CHW fp32,
// value / 255 (vae/training/train.py:158-179, 266-274 build that batch on the host with DataLoader workers).
#include "common.h"
#include "prof.h"
#include "../../include/vae_hip.h"

namespace vae {

// one thread per pixel: C consecutive source bytes, C plane stores (coalesced per plane)
template <int C>
__global__ __launch_bounds__(256) void feed_u8_kernel(const uint8_t* __restrict__ data, const long* __restrict__ index,
                                                      float* __restrict__ out, int hw, long n_items, long n_data) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += (long)gridDim.x * blockDim.x) {
        const long b = it / hw;
        const int p = (int)(it - b * hw);
        long src = index ? index[b] : b;
        if (src < 0) src += n_data;                        // negative indices count from the end, as in torch
        const bool ok = src >= 0 && src < n_data;          // out-of-range rows come out as zeros (the host checks when it can)
        const uint8_t* s = data + (src * hw + p) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) out[(b * C + c) * hw + p] = ok ? __fdiv_rn((float)s[c], 255.0f) : 0.f;
    }
}

}  // namespace vae

using namespace vae;

extern "C" int vae_feed_u8(const void* data_u8_nhwc, long n_data, const long* index, float* out_nchw, int batch, int C,
                           int H, int W, void* stream) {
    VAE_CHECK(data_u8_nhwc && out_nchw, "vae_feed_u8: null pointer");
    VAE_CHECK(n_data > 0 && batch > 0 && H > 0 && W > 0, "vae_feed_u8: bad sizes n=%ld batch=%d H=%d W=%d", n_data, batch, H, W);
    VAE_CHECK(C == 1 || C == 3 || C == 4, "vae_feed_u8: channels=%d not supported (1, 3 or 4)", C);
    VAE_CHECK(index || batch <= n_data, "vae_feed_u8: batch=%d exceeds the %ld stored images and no index was given", batch, n_data);
    const int hw = H * W;
    const long items = (long)batch * hw;
    long blocks = (items + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof("feed_u8_kernel", 0.0, (double)items * C * 5.0, s);
    const uint8_t* d = (const uint8_t*)data_u8_nhwc;
    if (C == 1) hipLaunchKernelGGL(feed_u8_kernel<1>, dim3((int)blocks), dim3(256), 0, s, d, index, out_nchw, hw, items, n_data);
    else if (C == 3) hipLaunchKernelGGL(feed_u8_kernel<3>, dim3((int)blocks), dim3(256), 0, s, d, index, out_nchw, hw, items, n_data);
    else hipLaunchKernelGGL(feed_u8_kernel<4>, dim3((int)blocks), dim3(256), 0, s, d, index, out_nchw, hw, items, n_data);
    VAE_LAUNCH_CHECK("feed_u8_kernel");
    return 0;
}
