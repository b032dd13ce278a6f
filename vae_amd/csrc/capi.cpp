// Version and thread-local error string of libvae_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/vae_hip.h"

namespace vae {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace vae

extern "C" int vae_version(void) { return VAE_HIP_VERSION; }
extern "C" const char* vae_last_error(void) { return vae::g_err; }
