import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "model_tflops", "model_mfma_frac", "final_loss", "kernel_ms_per_step_sum", "n_gpus")})
print(d.get("cpu_baseline"))
for r in d.get("kernels", []):
    print("%-36s n/step %5.1f ms/step %7.3f avg_us %8.1f %8.2f %-8s frac %.3f gbs %.0f" % (r["kernel"], r["launches_per_step"], r["ms_per_step"], r["avg_us"], r["achieved"], r["unit"], r["frac"], r["gbs"]))
