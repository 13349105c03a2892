#!/bin/bash
# Collects a round's evidence on a GPU box: the default bench line, kernel traces of the frame pipeline
# and of the big matcher search, the memory-traffic counter passes and the SQ / cache counter passes
# (each counter pass is a run of its own). Every GPU step runs under its own time limit; the script
# stops at the first step that fails.
# usage (from the repo root): bash tools/profile_round.sh r03 [steps...]   steps: bench trace match pmc sq surf 4k micro
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
PIPE="--no-cpu-baseline --no-verify --no-host-leg --no-two-streams-leg --no-configs --match-rows 0"
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
if has surf; then
    echo "[surf] kernel trace and counters of the SURF batch (64 frames 1080p BGR, BASELINE.json configs[1])"
    SURF="python $ROOT/tools/bench_surf.py --skip-single --batch 64 --steps 5"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_surf" -o t -- $SURF > "$OUT/surf_bench.json" 2> "$OUT/surf_trace.err"
    cp "$OUT"/trace_surf/t_kernel_stats.csv "$OUT/surf_kernel_stats.csv"
    python "$ROOT/tools/kstats.py" "$OUT/surf_kernel_stats.csv" 7 12
    timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES \
        --output-format csv -d "$OUT/pmc_surf_sq" -o q -- $SURF > "$OUT/pmc_surf_sq.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum \
        --output-format csv -d "$OUT/pmc_surf_mem" -o q -- $SURF > "$OUT/pmc_surf_mem.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_surf_fetch" -o q -- $SURF > "$OUT/pmc_surf_fetch.log" 2>&1
    timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_surf_write" -o q -- $SURF > "$OUT/pmc_surf_write.log" 2>&1
    (cd "$ROOT/tools" && python surf_traffic.py "$OUT"/pmc_surf_fetch/q_counter_collection.csv \
        "$OUT"/pmc_surf_write/q_counter_collection.csv 64 "$OUT/surf_traffic.json")
    for c in sq mem fetch write; do
        python "$ROOT/tools/pmc_summary.py" "$OUT"/pmc_surf_$c/q_counter_collection.csv | grep -v "rocclr\|at::native"
    done > "$OUT/surf_counters_batch64.txt"
    cat "$OUT/surf_counters_batch64.txt" | cut -c1-300
fi

if has 4k; then
    echo "[4k] kernel trace of the 4K single-channel batch (64 frames, extraction only, BASELINE.json configs[3])"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_4k" -o t -- \
        python "$ROOT/bench.py" $PIPE --width 3840 --height 2160 --channels 1 --no-match --batch 64 --steps 5 --warmup 2 > "$OUT/bench_4k.json" 2> "$OUT/trace_4k.err"
    cp "$OUT"/trace_4k/t_kernel_stats.csv "$OUT/4k_kernel_stats.csv"
    python "$ROOT/tools/kstats.py" "$OUT/4k_kernel_stats.csv" 7 14
    echo "[4k] FETCH_SIZE / WRITE_SIZE passes at batch 16"
    K4="--width 3840 --height 2160 --channels 1 --no-match --batch 16 --steps 2 --warmup 1"
    timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_4k_fetch" -o f -- \
        python "$ROOT/bench.py" $PIPE $K4 > "$OUT/pmc_4k_fetch.log" 2>&1
    timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_4k_write" -o w -- \
        python "$ROOT/bench.py" $PIPE $K4 > "$OUT/pmc_4k_write.log" 2>&1
    (cd "$ROOT" && python tools/pmc_traffic.py "$OUT"/pmc_4k_fetch/f_counter_collection.csv \
        "$OUT"/pmc_4k_write/w_counter_collection.csv 16 "$OUT/4k_blur_traffic.json")
fi

if has micro; then
    echo "[micro] vector issue rates and the host link"
    timeout -k 10 120 "$ROOT/tools/micro/valu_rate" > "$OUT/valu_rate.txt" 2>&1
    cat "$OUT/valu_rate.txt"
    timeout -k 10 120 "$ROOT/tools/micro/h2d_bw" > "$OUT/h2d_bw.txt" 2>&1
    tail -20 "$OUT/h2d_bw.txt"
fi
echo "[done] $STEPS"
