"""Fold two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE) over bench.py into
per-kernel HBM traffic per launch, keyed by the kernel names bench.py reports.

    python tools_pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units and corrections: rocprofv3 reports both counters in KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide (16 B per lane) coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for
16-byte-per-lane stores.  The result is bytes per launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE), averaged
over the launches of each kernel in the profiled steps.
"""
import collections
import csv
import json
import re
import sys


def bench_name(rocprof_name: str):
    """map a rocprofv3 kernel name onto the name the library's event profiler (and bench.py) uses"""
    n = rocprof_name
    m = re.search(r"conv3_patch_tn_kernel<\d+, (\d+), \d+(?:, (\d+))?>", n)
    if m:
        return f"conv3_patch_tn_kernel<{m.group(1)}" + (",ksplit>" if m.group(2) == "2" else ">")
    if "conv3_wide_kernel" in n:
        return "conv3_wide_kernel"
    if "conv3_ws64_kernel" in n:
        return "conv3_ws64_kernel"
    if "conv3_wgrad_patch_kernel" in n:
        return "conv3_wgrad_patch_kernel"
    m = re.search(r"gemm_tn_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), \d+>", n)
    if m:
        return "gemm_tn_kernel<" + ",".join(m.groups()) + ">"
    m = re.search(r"gemm_nt_kernel<(\d+), (\d+), (\d+), \d+>", n)
    if m:
        return "gemm_nt_kernel<" + ",".join(m.groups()) + ">"
    m = re.search(r"bn_act_bwd_kernel<(true|false), \d", n)
    if m:
        return "bn_act_bwd(reduce+apply)" + ("<pool>" if m.group(1) == "true" else "")
    m = re.search(r"bn_act_fwd_kernel<(true|false)>", n)
    if m:
        return "bn_act_fwd_kernel" + ("<pool>" if m.group(1) == "true" else "")
    m = re.search(r"vae::(\w+)", n)
    return m.group(1) if m else n


def fold(path, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = bench_name(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    fetch, nf = fold(sys.argv[1], "FETCH_SIZE")
    write, nw = fold(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in sorted(set(fetch) | set(write)):
        launches = max(nf.get(k, 0), nw.get(k, 0))
        if not launches:
            continue
        # the two-pass bn_act_bwd / head_bwd rows of bench.py cover two kernels per call: sum over both passes
        per_call = launches / 2 if k.startswith(("bn_act_bwd", "head_bwd")) else launches
        out[k] = {"launches_profiled": launches,
                  "fetch_bytes_per_launch": 1024.0 * 2.0 * fetch.get(k, 0.0) / per_call,
                  "write_bytes_per_launch": 1024.0 * write.get(k, 0.0) / per_call}
        out[k]["hbm_bytes_per_launch"] = out[k]["fetch_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    json.dump({"note": "bytes per launch = 1024 * (2 * FETCH_SIZE + WRITE_SIZE), rocprofv3 --pmc, separate passes; "
                       "FETCH_SIZE doubled for gfx950's 64-byte tally of 128-byte requests",
               "workload": f"python3 bench.py --config {sys.argv[4] if len(sys.argv) > 4 else 'cfg2_1gpu'} --steps 5 --warmup 2", "kernels": out},
              open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k:40s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch (read {v['fetch_bytes_per_launch'] / 1e6:7.1f}, "
              f"write {v['write_bytes_per_launch'] / 1e6:7.1f})")


if __name__ == "__main__":
    main()
