#!/bin/bash
# A/B of library builds inside ONE gpurun call (same box, same clocks):
#   gpurun -- 'bash tools/exp_run.sh [-b "<bench flags>"] default cand default cand'
# `default` is the in-tree library, any other name is gpurun_exp/<name>.so (tools/build_exp.sh). Every
# variant runs the resident-frame headline without the checker and the other legs and prints its
# frames/s and stage times; the JSON lines stay under gpurun_out/exp/.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD
O=$R/gpurun_out/exp
mkdir -p "$O"
FLAGS="--steps 10"
if [ "$1" = "-b" ]; then FLAGS=$2; shift 2; fi
AB="python3 $R/bench.py --no-cpu-baseline --no-verify --no-host-leg --no-two-streams-leg --no-configs --match-rows 0 $FLAGS"
i=0
for v in "$@"; do
    i=$((i + 1))
    if [ "$v" = default ]; then
        timeout -k 10 300 $AB > "$O/ab_${i}_$v.json" 2> "$O/ab_${i}_$v.err" || exit 1
    else
        MIF_LIBRARY=$R/gpurun_exp/$v.so timeout -k 10 300 $AB > "$O/ab_${i}_$v.json" 2> "$O/ab_${i}_$v.err" || exit 1
    fi
    python3 -c "import json; d = json.load(open('$O/ab_${i}_$v.json')); print('$v', round(d['value'], 1), d['stage_ms_per_batch'])"
done
