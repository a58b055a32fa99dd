#!/bin/bash
# rocprofv3 PMC passes for the L1 / L2 side of the bench frame's kernels (run on the GPU box from the repo root):
#   tools/pmc_l2.sh <out_dir>      -> <out_dir>/l2_pmc.txt   (per kernel and grid: requests, hit rate, L1->L2 read requests and their mean latency)
# Own passes, counters only (never together with --kernel-trace / --stats).  Why: DESIGN.md 3b prices the 3x3 strip kernel's weight streaming
# (every workgroup reads the whole [128 couts][9 Cin] weight slab from L2) from the algorithm; this measures it.
set -u
OUT=${1:-gpurun_out/pmc_l2}
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-extra --no-ingest --frames-in-flight 1"
cd /tmp
pass() {
    local name=$1; shift
    timeout 600 rocprofv3 --pmc "$@" -d "$ROOT/$OUT/$name" -o p -- $CMD > "$ROOT/$OUT/$name.log" 2>&1
    echo "pass $name rc=$?"
}
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
cd "$ROOT"
python - "$OUT" > "$OUT/l2_pmc.txt" <<'PY'
import collections, glob, os, re, sqlite3, sys
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for db in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
    cur = sqlite3.connect(db).cursor()
    for did, name, grid, wg, cname, val in cur.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection"):
        key = (re.sub(r"cobevt::|void |\(anonymous namespace\)::", "", name)[:70], grid // max(wg, 1))
        res[key][cname][0] += val
        res[key][cname][1] += 1
print("# per kernel launch (means over the traced launches): L2 requests / hit rate (TCC_*), vector-L1 -> L2 read requests and their mean latency in cycles (TCP_TCC_READ_REQ*),")
print("# texture-addresser busy cycles summed over the CUs' TAs (TA_TA_BUSY_sum) against 256 x GRBM_GUI_ACTIVE")
for (name, wgs), c in sorted(res.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", [0, 1])[0]):
    m = lambda k: c[k][0] / max(c[k][1], 1) if k in c else float("nan")
    req, hit, miss = m("TCC_REQ_sum"), m("TCC_HIT_sum"), m("TCC_MISS_sum")
    rd, lat, ta, act = m("TCP_TCC_READ_REQ_sum"), m("TCP_TCC_READ_REQ_LATENCY_sum"), m("TA_TA_BUSY_sum"), m("GRBM_GUI_ACTIVE")
    if not (req == req) or req < 1e4:
        continue
    print("%-72s %5d wgs  TCC req %.3g hit %.3g miss %.3g (hit rate %.3f) | L1->L2 reads %.3g, mean latency %.0f cyc | TA busy %.3g = %.2f of 256 x %.3g active cycles"
          % (name, wgs, req, hit, miss, hit / max(hit + miss, 1), rd, lat / max(rd, 1), ta, ta / max(256 * act, 1), act))
PY
rm -rf "$OUT/tcc" "$OUT/tcp"
head -30 "$OUT/l2_pmc.txt" | cut -c1-260
