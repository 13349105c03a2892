#!/bin/bash
# Collects a round's evidence on a GPU box: the default bench line, kernel traces of the frame pipeline
# and of the big matcher search, the memory-traffic counter passes and the SQ / cache counter passes
# (each counter pass is a run of its own). Every GPU step runs under its own time limit; the script
# stops at the first step that fails.
# usage (from the repo root): bash tools/profile_round.sh r02 [steps...]   steps: bench trace match pmc sq
# Writes under gpurun_out/<tag>/; copy what is to be kept into profiles/.
set -e -o pipefail
TAG=${1:-r02}
shift || true
STEPS=${*:-bench trace match pmc sq}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# the frame pipeline alone: no matcher block, no host-to-host leg, no CPU legs
PIPE="--no-cpu-baseline --no-verify --no-host-leg --no-two-streams-leg --match-rows 0"
has() { [[ " $STEPS " == *" $1 "* ]]; }

if has bench; then
    echo "[bench] default configuration"
    timeout -k 10 600 python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
    tail -c 600 "$OUT/bench.json"; echo
fi

if has trace; then
    echo "[trace] kernel trace of the frame pipeline"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
        python "$ROOT/bench.py" $PIPE --steps 5 --warmup 2 > "$OUT/trace.log" 2>&1
    cp "$OUT"/trace/t_kernel_stats.csv "$OUT/bench_kernel_stats.csv"
    python "$ROOT/tools/kstats.py" "$OUT/bench_kernel_stats.csv" 7 24
fi

if has match; then
    echo "[match] kernel trace of the 1 M x 1 M search, screened and plain"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_m" -o s -- \
        python "$ROOT/tools/bench_match.py" --nq 1000000 --nd 1000000 --steps 2 --warmup 1 > "$OUT/match_screened.json" 2> "$OUT/match_screened.err"
    cp "$OUT"/trace_m/s_kernel_stats.csv "$OUT/match_screened_kernel_stats.csv"
    MIF_MATCH_NO_SCREEN=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_m" -o e -- \
        python "$ROOT/tools/bench_match.py" --nq 1000000 --nd 1000000 --steps 2 --warmup 1 > "$OUT/match_exact.json" 2> "$OUT/match_exact.err"
    cp "$OUT"/trace_m/e_kernel_stats.csv "$OUT/match_exact_kernel_stats.csv"
    cat "$OUT/match_screened.json" "$OUT/match_exact.json"
fi

if has pmc; then
    echo "[pmc] FETCH_SIZE pass"
    timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- \
        python "$ROOT/bench.py" $PIPE --batch 64 --steps 2 --warmup 1 > "$OUT/pmc_fetch.log" 2>&1
    echo "[pmc] WRITE_SIZE pass"
    timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- \
        python "$ROOT/bench.py" $PIPE --batch 64 --steps 2 --warmup 1 > "$OUT/pmc_write.log" 2>&1
    (cd "$ROOT" && python tools/pmc_traffic.py "$OUT"/pmc_fetch/f_counter_collection.csv \
        "$OUT"/pmc_write/w_counter_collection.csv 64 "$OUT/blur_traffic.json")
fi

if has sq; then
    echo "[sq] wave-cycle counters at the bench batch"
    timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
        --output-format csv -d "$OUT/pmc_sq" -o q -- \
        python "$ROOT/bench.py" $PIPE --batch 512 --steps 1 --warmup 1 > "$OUT/pmc_sq.log" 2>&1
    python "$ROOT/tools/pmc_summary.py" "$OUT"/pmc_sq/q_counter_collection.csv > "$OUT/sq_counters_batch512.txt"
    echo "[sq] instruction counters at the bench batch"
    timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT \
        --output-format csv -d "$OUT/pmc_sq2" -o q -- \
        python "$ROOT/bench.py" $PIPE --batch 512 --steps 1 --warmup 1 > "$OUT/pmc_sq2.log" 2>&1
    python "$ROOT/tools/pmc_summary.py" "$OUT"/pmc_sq2/q_counter_collection.csv > "$OUT/sq_insts_batch512.txt"
    echo "[sq] L2 and vector-cache counters at the bench batch"
    timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum \
        --output-format csv -d "$OUT/pmc_l2" -o q -- \
        python "$ROOT/bench.py" $PIPE --batch 512 --steps 1 --warmup 1 > "$OUT/pmc_l2.log" 2>&1
    python "$ROOT/tools/pmc_summary.py" "$OUT"/pmc_l2/q_counter_collection.csv > "$OUT/l2_counters_batch512.txt"
fi
echo "[done] $STEPS"
