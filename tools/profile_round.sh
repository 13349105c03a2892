#!/bin/bash
# Collects the round's evidence on a GPU box: the default bench line, a kernel trace of the same
# command and the two memory-traffic counter passes (each counter pass is a run of its own).
# usage (from the repo root): bash tools/profile_round.sh r01
# Writes under gpurun_out/<tag>/; copy what is to be kept into profiles/.
set -e -o pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp

echo "[1/5] bench (default configuration)"
timeout -k 10 420 python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 400 "$OUT/bench.json"; echo

echo "[2/5] kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
    python "$ROOT/bench.py" --no-cpu-baseline --steps 5 --warmup 2 > "$OUT/trace.log" 2>&1
cp "$OUT"/trace/t_kernel_stats.csv "$OUT/bench_kernel_stats.csv"

echo "[3/5] FETCH_SIZE pass"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- \
    python "$ROOT/bench.py" --no-cpu-baseline --batch 64 --steps 2 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1

echo "[4/5] WRITE_SIZE pass"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- \
    python "$ROOT/bench.py" --no-cpu-baseline --batch 64 --steps 2 --warmup 1 > "$OUT/pmc_write.log" 2>&1

echo "[5/5] traffic summary"
cd "$ROOT"
python tools/pmc_traffic.py "$OUT"/pmc_fetch/f_counter_collection.csv "$OUT"/pmc_write/w_counter_collection.csv 64 \
    "$OUT/blur_traffic.json"
python tools/kstats.py "$OUT/bench_kernel_stats.csv" 7 24

echo "[extra] SQ counters of the extraction and matcher kernels (batch 16)"
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/pmc_sq" -o q -- \
    python "$ROOT/bench.py" --no-cpu-baseline --batch 16 --steps 2 --warmup 1 > "$OUT/pmc_sq.log" 2>&1
cd "$ROOT"
python tools/pmc_summary.py "$OUT"/pmc_sq/q_counter_collection.csv > "$OUT/sq_counters_batch16.txt"
