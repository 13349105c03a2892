#!/bin/bash
# A/B of library builds on the SURF batch (tools/bench_surf.py, 64 1080p frames) inside ONE gpurun call:
#   gpurun -- 'bash tools/exp_surf.sh default cand default cand'   (names as in tools/exp_run.sh)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD
O=$R/gpurun_out/exp
mkdir -p "$O"
i=0
for v in "$@"; do
    i=$((i + 1))
    if [ "$v" = default ]; then L=; else L=$R/gpurun_exp/$v.so; fi
    MIF_LIBRARY=$L timeout -k 10 200 python3 $R/tools/bench_surf.py --skip-single --batch 64 --steps 20 > "$O/s_${i}_$v.json" 2> "$O/s_${i}_$v.err" || exit 1
    python3 -c "import json; d = json.load(open('$O/s_${i}_$v.json'))['batched']; print('$v', round(d['frames_per_s']), round(d['ms_per_batch'], 3))"
done
