#!/bin/bash
# Regenerates the judged summaries under gpurun_out/refresh (run on the GPU box from the repo root), to be copied into profiles/:
#   PMC passes -> pmc_<tag>.json (+ the LDS pass), the default bench line (reads the fresh PMC file), the kernel-trace summary of
#   the default command and the one-frame timeline.  The rocprofv3 .db directories are deleted before the job ends (64 MiB cap).
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=gpurun_out/refresh
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/pmc_collect.sh "$OUT/pmc" "$TAG" > "$OUT/pmc_collect.log" 2>&1
cp "$OUT/pmc/pmc_${TAG}.json" "profiles/pmc_${TAG}.json" 2>/dev/null       # bench.py reads roofline.traffic from it
bash tools/pmc_collect.sh "$OUT/pmc_lidar" "$TAG" lidar > "$OUT/pmc_collect_lidar.log" 2>&1      # the LiDAR FuseBEVT workload's own PMC passes
cp "$OUT/pmc_lidar/pmc_lidar_${TAG}.json" "profiles/pmc_lidar_${TAG}.json" 2>/dev/null
rm -rf "$OUT"/pmc_lidar/fetch "$OUT"/pmc_lidar/write "$OUT"/pmc_lidar/sq
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
    -d "$ROOT/$OUT/lds" -o p -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extra --frames-in-flight 1 > "$ROOT/$OUT/lds.log" 2>&1)
python tools/pmc_summary.py $(find "$OUT/lds" -name "*.db") > "$OUT/${TAG}_lds_pmc.txt" 2>&1
timeout 600 python bench.py > "$OUT/${TAG}_bench_full.json" 2> "$OUT/bench_full.err"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/kt" -o p -- python "$ROOT/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-extra --no-ingest > "$ROOT/$OUT/kt.log" 2>&1)
python tools/rocprof_summary.py $(find "$OUT/kt" -name "*.db" | head -1) --steady 30 --skip-last 55 --busy 10 > "$OUT/${TAG}_kernel_trace_stats.txt" 2>&1
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d "$ROOT/$OUT/tl" -o p -- python "$ROOT/bench.py" --steps 30 --warmup 10 --frames-in-flight 1 --no-cpu-baseline --no-roofline --no-extra --no-ingest > "$ROOT/$OUT/tl.log" 2>&1)
python tools/rocprof_summary.py $(find "$OUT/tl" -name "*.db" | head -1) --steady 20 --by-grid --timeline 80 --densest > "$OUT/${TAG}_frame_timeline.txt" 2>&1
rm -rf "$OUT/kt" "$OUT/tl" "$OUT/lds" "$OUT"/pmc/fetch "$OUT"/pmc/write "$OUT"/pmc/sq
ls -la "$OUT"
cut -c1-400 "$OUT/${TAG}_bench_full.json"
