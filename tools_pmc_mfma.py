"""Fold one rocprofv3 pass (--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace)
over bench.py into the matrix-pipe utilisation of every kernel that issues MFMAs.

    python tools_pmc_mfma.py <counter_collection.csv> <kernel_trace.csv> <out.json>

SQ_VALU_MFMA_BUSY_CYCLES is the sum over all SIMDs of the cycles their matrix pipe was busy: 16 per
v_mfma_f32_16x16x32_bf16 (checked against the 64->64 conv at 32x32, batch 512: 2 359 296 MFMAs, counter
37 748 736).  Utilisation = busy cycles / (1024 SIMDs x kernel duration x 2.4 GHz); with 1024 flop per SIMD
per cycle that is the same quantity as achieved / peak dense bf16 FLOP/s, read from the hardware instead of
from the layer shapes.  Durations are those of the profiled run (counters slow short kernels down a little).
"""
import collections
import csv
import json
import sys

CLOCK_GHZ, SIMDS = 2.4, 1024


def main():
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[2]))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sys.argv[1])):
        name = r["Kernel_Name"].split("(")[0].split("vae::")[-1]
        a = agg[name]
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a["launches"] += 1
            a["ns"] += dur.get(r["Dispatch_Id"], 0)
    out = {}
    for k, a in agg.items():
        if a["SQ_VALU_MFMA_BUSY_CYCLES"] <= 0 or a["ns"] <= 0:
            continue
        out[k] = {"launches_profiled": int(a["launches"]), "avg_us": a["ns"] / a["launches"] / 1e3,
                  "mfma_busy_cycles_per_launch": a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["launches"],
                  "mfma_util": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * a["ns"] * CLOCK_GHZ),
                  "tflops": a["SQ_VALU_MFMA_BUSY_CYCLES"] / 16 * 16384 / a["ns"] / 1e3}
    json.dump({"note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * duration * 2.4 GHz); 16 busy cycles per "
                       "v_mfma_f32_16x16x32_bf16; tflops = MFMAs * 16384 flop / duration (padded tiles included)",
               "workload": f"python3 bench.py --config {sys.argv[4] if len(sys.argv) > 4 else 'cfg2_1gpu'} --steps 3 --warmup 2", "kernels": out},
              open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_profiled"]):
        print(f"{k:52s} n={v['launches_profiled']:3d} {v['avg_us']:7.1f} us  util {100 * v['mfma_util']:5.1f} %  {v['tflops']:6.0f} TF")


if __name__ == "__main__":
    main()
