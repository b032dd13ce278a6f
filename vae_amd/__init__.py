"""MI355X-native conv-VAE hot path (HIP kernels behind a C ABI)."""
