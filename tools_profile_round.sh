#!/bin/bash
# Evidence passes of one benchmark config, run on the GPU box:  bash tools_profile_round.sh <config> <tag>
#   <config>  a yaml under config/ (cfg2_1gpu, cfg4_celeba64, ...);  <tag>  suffix of the files left in gpurun_out/prof_<tag>/
# One rocprofv3 run per pass (counters never share a run with tracing beyond --kernel-trace; FETCH_SIZE and WRITE_SIZE
# cannot share one): kernel trace + stats, matrix-pipe counters, fetch bytes, write bytes.  Each pass has its own time limit
# and the script stops at the first failure.
set -e
CFG=${1:-cfg2_1gpu}; TAG=${2:-cfg2}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
B="python3 bench.py --config $CFG --no-cpu-baseline --no-secondary"
cd "$PWD"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- $B --steps 30 --warmup 5 > "$OUT/stats.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/mfma" -o run -- $B --steps 3 --warmup 2 --no-kernel-profile > "$OUT/mfma.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o run -- $B --steps 5 --warmup 2 --no-kernel-profile > "$OUT/fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o run -- $B --steps 5 --warmup 2 --no-kernel-profile > "$OUT/write.log" 2>&1
f() { find "$OUT/$1" -name "*$2" | head -1; }
python3 tools_pmc_mfma.py "$(f mfma counter_collection.csv)" "$(f mfma kernel_trace.csv)" "$OUT/pmc_mfma.json" "$CFG" > "$OUT/pmc_mfma.txt"
python3 tools_pmc_traffic.py "$(f fetch counter_collection.csv)" "$(f write counter_collection.csv)" "$OUT/pmc_traffic.json" "$CFG" > "$OUT/pmc_traffic.txt"
cp "$(f stats kernel_stats.csv)" "$OUT/kernel_stats.csv"
# keep the folded results only (the raw traces are tens of MB)
rm -rf "$OUT/stats" "$OUT/mfma" "$OUT/fetch" "$OUT/write"
echo done
