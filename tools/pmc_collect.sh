#!/bin/bash
# rocprofv3 PMC passes over eager frames of the bench workload (run on the GPU box from the repo root):
#   tools/pmc_collect.sh <out_dir> [tag] [camera|lidar]     (lidar: bench.py --workload lidar -> pmc_lidar_<tag>.json)
# Separate passes as MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2; SQ counters in their
# own pass; never together with --kernel-trace / --stats).  Summarise with tools/pmc_roofline.py.
set -u
OUT=${1:-gpurun_out/pmc}
TAG=${2:-r02}
WORKLOAD=${3:-camera}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extra --no-ingest --frames-in-flight 1"
NAME="pmc_${TAG}.json"
if [ "$WORKLOAD" = "lidar" ]; then
    CMD="python $ROOT/bench.py --workload lidar --steps 3 --warmup 1 --no-graph --no-roofline"
    NAME="pmc_lidar_${TAG}.json"
fi
cd /tmp
rocprofv3 -L > "$ROOT/$OUT/counters_available.txt" 2>&1 || true
pass() {   # name, counters...
    local name=$1; shift
    timeout 600 rocprofv3 --pmc "$@" -d "$ROOT/$OUT/$name" -o p -- $CMD > "$ROOT/$OUT/$name.log" 2>&1
    echo "pass $name rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
if ! ls "$ROOT/$OUT/sq"/*.db > /dev/null 2>&1; then          # an unknown counter name aborts the pass: retry with the core set
    pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE
fi
cd "$ROOT"
python tools/pmc_roofline.py "$OUT" > "$OUT/$NAME" && echo "wrote $OUT/$NAME"
