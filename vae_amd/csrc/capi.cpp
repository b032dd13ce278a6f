// Version, thread-local error string, and the optional per-kernel event profiler of
// libvae_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vae_hip.h"
#include "prof.h"

namespace vae {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- event profiler
// Off by default (one branch per launch).  When armed by vae_prof_start, every instrumented
// launch is bracketed by a pair of pre-created hipEvents on the launch stream; vae_prof_stop
// synchronises, reads the elapsed times and aggregates them per kernel name together with
// the algorithmic FLOPs / bytes the launcher declared.  Events are created in
// vae_prof_start (never inside a launch), so the launch path itself allocates nothing.
struct ProfRecord {
    const char* name;
    double flops, bytes;
};
// The profiler is one process-wide table behind a mutex; the only thing a launch reads when it is off is the
// atomic flag.  It is a single-timeline tool: while it is armed, launches are expected from ONE host thread
// (begin/end pairs of two threads would interleave in the table), which is how bench.py uses it.
static std::mutex g_mu;
static std::vector<hipEvent_t> g_ev;
static std::vector<ProfRecord> g_rec;
static std::atomic<bool> g_on{false};
static size_t g_cap = 0;

bool prof_enabled() { return g_on.load(std::memory_order_acquire); }

void prof_begin(const char* name, double flops, double bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_on || g_rec.size() >= g_cap) return;
    g_rec.push_back({name, flops, bytes});
    (void)hipEventRecord(g_ev[2 * (g_rec.size() - 1)], s);   // a failed record only loses this sample
}
void prof_end(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_on || g_rec.empty() || g_rec.size() > g_cap) return;
    (void)hipEventRecord(g_ev[2 * (g_rec.size() - 1) + 1], s);
}
}  // namespace vae

namespace vae {
extern int g_conv3_wide_mode;
extern int g_conv3_ksplit_mode;
extern int g_stream_gemm_mode;
extern int g_wgrad_pc_mode;
extern int g_conv3_ws64_mode;
extern int g_conv1_bwd_mode;
extern int g_pool_ymax_mode;
extern int g_head_sums_mode;
extern int g_fold_defer_mode;
extern int g_slab_reparam_mode;
extern int g_convt_bwd_mode;
extern int g_wgrad_batch_mode;
extern int g_reduce_direct_mode;
extern int g_conv1_twice_mode;
extern int g_wgrad_ablate;
extern unsigned long long* g_conv3_dbg;
}
// Diagnostic aid: a device buffer that diagnostic kernel variants write cycle stamps to (null = none).
extern "C" int vae_set_debug_buffer(void* dev_ptr) {
    vae::g_conv3_dbg = (unsigned long long*)dev_ptr;
    return 0;
}
// Measurement aid: selects between kernel variants that compute the same result, for A/B timing in one process.
// Every key has a closed range; a value outside it is rejected instead of being stored.
extern "C" int vae_set_tuning(const char* key, int value) {
    struct Key {
        const char* name;
        int* slot;
        int lo, hi;
    };
    static const Key keys[] = {
        {"conv3_wide", &vae::g_conv3_wide_mode, 0, 2},        // 0 auto, 1 always the 4-wave kernel, 2 the 8-wave kernel wherever legal
        {"conv3_ksplit", &vae::g_conv3_ksplit_mode, 0, 2},    // 0 auto, 1 never split K inside the workgroup, 2 wherever legal
        {"wgrad_pc", &vae::g_wgrad_pc_mode, 0, 2},            // conv3x3 weight gradient, copying and computing waves separated: 0 by K range, 1 always, 2 never
        {"conv3_ws64", &vae::g_conv3_ws64_mode, 0, 4},        // weight-stationary conv3x3 for 64 input channels: 0 where its jobs fill the chip, 1 never, 2 wherever legal, 3 as 0, hand-pipelined
        {"conv1_bwd", &vae::g_conv1_bwd_mode, 0, 1},          // first conv's BatchNorm backward + weight gradient: 0 the one-pass kernel where Cin <= 3, 1 separate passes
        {"pool_ymax", &vae::g_pool_ymax_mode, 0, 1},          // pooled BatchNorm backward sums from the stored window maxima: 0 yes, 1 re-read the windows
        {"head_sums", &vae::g_head_sums_mode, 0, 1},          // the 1x1 head's backward sums: 0 on the matrix pipe, 1 in vector registers
        {"fold_defer", &vae::g_fold_defer_mode, 0, 1},        // whole-pass backward: 0 one slab-fold launch at the end, 1 one per stage
        {"slab_reparam", &vae::g_slab_reparam_mode, 0, 1},    // split-K slabs of the latent GEMMs: 0 added up by the reparameterisation kernels, 1 by a finishing kernel
        {"convt_bwd", &vae::g_convt_bwd_mode, 0, 2},          // 64 -> 64 ConvTranspose backward: 0 one kernel that also makes the head's gradient, 1 separate kernels, 2 one kernel reading the stored gradient
        {"wgrad_ablate", &vae::g_wgrad_ablate, 0, 7},         // diagnostic builds (-DVAE_ABLATE) only: bit 0 no matrix instructions, 1 no copies, 2 no fragment reads (32-wide images)
        {"conv1_twice", &vae::g_conv1_twice_mode, 0, 1},      // first conv: 0 convolved twice, its output never stored (needs conv1_bwd = 0), 1 stored
        {"reduce_direct", &vae::g_reduce_direct_mode, 0, 1},  // slab folds of at most 8 slabs: 1 one thread per float4 column, no barrier (default), 0 the split-group form
        {"wgrad_batch", &vae::g_wgrad_batch_mode, 0, 1},      // conv3x3 weight gradients of a backward stage: 0 ONE launch for all of them, 1 one launch per layer
        {"stream_gemm", &vae::g_stream_gemm_mode, 0, 2},      // 0 the streaming GEMM where it pays, 1 never (tiled GEMM), 2 wherever legal
    };
    for (const Key& k : keys) {
        if (key && !strcmp(key, k.name)) {
            if (value < k.lo || value > k.hi) {
                vae::set_error("vae_set_tuning: %s=%d out of range [%d, %d]", k.name, value, k.lo, k.hi);
                return 2;
            }
            *k.slot = value;
            return 0;
        }
    }
    vae::set_error("vae_set_tuning: unknown key '%s'", key ? key : "(null)");
    return 1;
}

extern "C" int vae_version(void) { return VAE_HIP_VERSION; }
extern "C" const char* vae_last_error(void) { return vae::g_err; }

extern "C" int vae_prof_start(int max_launches) {
    using namespace vae;
    if (max_launches <= 0 || max_launches > (1 << 20)) {
        set_error("vae_prof_start: max_launches=%d out of range", max_launches);
        return 1;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    while (g_ev.size() < (size_t)2 * max_launches) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            set_error("vae_prof_start: hipEventCreate failed");
            return 2;
        }
        g_ev.push_back(e);
    }
    g_rec.clear();
    g_rec.reserve(max_launches);
    g_cap = max_launches;
    g_on = true;
    return 0;
}

// Writes up to max_rows aggregated rows; returns the number of distinct kernels (or -1).
// Row i: names[i] (64 bytes each), launches[i], total_ms[i], flops[i], bytes[i] (sums).
extern "C" int vae_prof_stop(char* names, int* launches, double* total_ms, double* flops, double* bytes,
                             int max_rows) {
    using namespace vae;
    std::lock_guard<std::mutex> lock(g_mu);
    g_on = false;
    if (hipDeviceSynchronize() != hipSuccess) {
        set_error("vae_prof_stop: hipDeviceSynchronize failed");
        return -1;
    }
    std::map<std::string, int> index;
    std::vector<std::string> order;
    std::vector<int> cnt;
    std::vector<double> ms, fl, by;
    for (size_t i = 0; i < g_rec.size(); ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_ev[2 * i], g_ev[2 * i + 1]) != hipSuccess) continue;
        auto it = index.find(g_rec[i].name);
        int k;
        if (it == index.end()) {
            k = (int)order.size();
            index[g_rec[i].name] = k;
            order.push_back(g_rec[i].name);
            cnt.push_back(0); ms.push_back(0); fl.push_back(0); by.push_back(0);
        } else {
            k = it->second;
        }
        cnt[k] += 1; ms[k] += t; fl[k] += g_rec[i].flops; by[k] += g_rec[i].bytes;
    }
    g_rec.clear();
    const int n = (int)order.size();
    for (int k = 0; k < n && k < max_rows; ++k) {
        if (names) {
            strncpy(names + 64 * k, order[k].c_str(), 63);
            names[64 * k + 63] = 0;
        }
        if (launches) launches[k] = cnt[k];
        if (total_ms) total_ms[k] = ms[k];
        if (flops) flops[k] = fl[k];
        if (bytes) bytes[k] = by[k];
    }
    return n;
}
