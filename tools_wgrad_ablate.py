"""What bounds the producer / consumer form of the conv3x3 weight-gradient kernel: the same launch with its matrix
instructions, its global->LDS copies or its LDS fragment reads left out (a library built with -DVAE_ABLATE).

    make -C vae_amd/csrc clean && make -C vae_amd/csrc -j8 EXTRA=-DVAE_ABLATE     # diagnostic build
    python tools_wgrad_ablate.py [out.json]                                          # on the GPU box
"""
import json
import sys

import torch

from vae_amd import _lib as L

DEV = "cuda"


def time_us(fn, iters=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wgrad_ablate.json"
    L.load()
    n, H, Ci, Co = 512, 32, 64, 64
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(n, H, H, Ci, device=DEV, generator=g).to(torch.bfloat16)
    dy = torch.randn(n, H, H, Co, device=DEV, generator=g).to(torch.bfloat16)
    rec = {"what": "conv3x3 weight gradient, 512 x 32x32, 64 -> 64 channels, producer / consumer form, us per launch",
           "rows": []}
    L.call("vae_set_tuning", b"wgrad_pc", 1)
    for ns in (256, 102):                       # 32 and 81 K tiles per workgroup (one layer alone / its share of a shared launch)
        slab = torch.empty(ns, Co, 9, Ci, device=DEV)
        row = {"nsplit": ns, "k_tiles_per_workgroup": -(-n * H * H // 64 // ns)}
        for abl, name in [(0, "full"), (1, "no_mfma"), (2, "no_copies"), (4, "no_frag_reads"), (3, "no_mfma_no_copies"),
                          (5, "no_mfma_no_reads"), (6, "no_copies_no_reads"), (7, "barriers_and_slab_store_only")]:
            L.call("vae_set_tuning", b"wgrad_ablate", abl)
            row[name] = round(time_us(lambda: L.call("vae_conv3_wgrad_patch", L.ptr(dy), L.ptr(x), L.ptr(slab), n, H, H, Ci, Co, ns,
                                                     L.current_stream())), 2)
        L.call("vae_set_tuning", b"wgrad_ablate", 0)
        rec["rows"].append(row)
        print(row, flush=True)
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
