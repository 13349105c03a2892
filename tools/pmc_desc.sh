#!/bin/bash
# SQ counters of the descriptor kernel for its two forms (lane-by-lane walk, lock step); one rocprofv3 pass each.
# usage (repo root, on a GPU box): bash tools/pmc_desc.sh <outdir>
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/${1:-gpurun_out/pmc_desc}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
PIPE="--no-cpu-baseline --no-verify --no-host-leg --match-rows 0 --batch 256 --steps 1 --warmup 1"
CNT="SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA"
for mode in 0 1; do
    export MIF_DESC_LOCKSTEP=$mode
    timeout -s KILL 240 rocprofv3 --pmc $CNT --output-format csv -d "$OUT/m$mode" -o q -- \
        python "$ROOT/bench.py" $PIPE > "$OUT/m$mode.log" 2>&1
    python "$ROOT/tools/pmc_summary.py" "$OUT/m$mode/q_counter_collection.csv" k_descriptor | tee "$OUT/desc_mode$mode.txt"
done
