// Optional per-kernel event profiler (see capi.cpp).  Launchers bracket their launch with
// prof_begin / prof_end; both are a single predictable branch when profiling is off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

namespace vae {
bool prof_enabled();
void prof_begin(const char* name, double flops, double bytes, hipStream_t s);
void prof_end(hipStream_t s);

struct ProfScope {
    hipStream_t s;
    bool on;
    ProfScope(const char* name, double flops, double bytes, hipStream_t st) : s(st), on(prof_enabled()) {
        if (on) prof_begin(name, flops, bytes, s);
    }
    ~ProfScope() {
        if (on) prof_end(s);
    }
};
}  // namespace vae
