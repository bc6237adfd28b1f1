#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r01
# Outputs under gpurun_out/<round>/; tools/summarise_profiles.py turns them into profiles/<round>_*.
set -o pipefail
step() { echo "[$(date +%T)] $*"; }
R=${1:-r01}
OUT=gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
# 1. the plain default bench line (what the driver runs)
timeout -k 10 500 python bench.py > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.err" ; step "command 1 rc=$?"
# 2. kernel trace + stats of the same command
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python bench.py > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err" ; step "command 2 rc=$?"
# 3. counters, each in its own pass (no tracing alongside), short run without the CPU baseline
ARGS="bench.py --steps 20 --warmup 2 --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python $ARGS > "$OUT/pmc_fetch.log" 2>&1 ; step "command 3 rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python $ARGS > "$OUT/pmc_write.log" 2>&1 ; step "command 4 rc=$?"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- python $ARGS > "$OUT/pmc_sq.log" 2>&1 ; step "command 5 rc=$?"
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_grbm" -- python $ARGS > "$OUT/pmc_grbm.log" 2>&1 ; step "command 6 rc=$?"
# 4. the C5 batch (both driving modes) and the sequential estimators under the kernel trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_trace" -- python bench.py --workload c5 --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/c5_under_rocprof.json" 2> "$OUT/c5_trace.err" ; step "command 7 rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/usac_trace" -- python tools/usac_profile_run.py 40 > "$OUT/usac_profile_run.txt" 2> "$OUT/usac_trace.err" ; step "command 8 rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5_usac_trace" -- python tools/c5_usac_timing.py 512 > "$OUT/c5_usac_under_rocprof.json" 2> "$OUT/c5_usac_trace.err" ; step "command 9 rc=$?"
# 5. summarise on the box (gpurun merges at most 64 MiB back; the raw per-dispatch tables of a default bench run are larger) and keep the
#    raw files small: no databases, no per-dispatch traces
MLPL_PROFILE_DST="gpurun_out/${R}_summary" python tools/summarise_profiles.py "$R" > "$OUT/summarise.log" 2>&1
find "$OUT" -name '*.db' -delete
find "$OUT" -name '*kernel_trace.csv' -delete
find "$OUT" -name '*counter_collection.csv' -delete
find "$OUT" -name '*agent_info.csv' -delete
du -sh "$OUT" "gpurun_out/${R}_summary"
