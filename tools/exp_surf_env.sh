#!/bin/bash
# The SURF batch (tools/bench_surf.py) under several settings of ONE environment switch, in one gpurun call:
#   gpurun -- 'bash tools/exp_surf_env.sh MIF_SURF_INTEGRAL twopass fused twopass fused'
# Optional first argument -b "<bench_surf flags>" (default: --skip-single --batch 64 --steps 20).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/exp
FLAGS="--skip-single --batch 64 --steps 20"
if [ "$1" = "-b" ]; then FLAGS=$2; shift 2; fi
NAME=$1; shift
i=0
for v in "$@"; do
    i=$((i + 1))
    if [ -z "$v" ]; then unset "$NAME"; else export "$NAME=$v"; fi
    timeout -k 10 200 python3 tools/bench_surf.py $FLAGS > gpurun_out/exp/senv_$i.json 2> gpurun_out/exp/senv_$i.err || { tail -3 gpurun_out/exp/senv_$i.err; exit 1; }
    python3 -c "import json; d = json.load(open('gpurun_out/exp/senv_$i.json')); print('$NAME=$v', {k: (round(v['frames_per_s']), round(v.get('ms_per_batch', 0), 3)) for k, v in d.items() if isinstance(v, dict) and 'frames_per_s' in v})"
done
