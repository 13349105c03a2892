#!/bin/bash
# The frame pipeline under several settings of ONE environment switch, in one gpurun call:
#   gpurun -- 'bash tools/exp_env.sh MIF_MATCH_SAMPLE "" 2 3 4'      ("" = switch not set)
# Every run is checked against the CPU checker (no --no-verify) and prints frames/s and stage times.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out/exp
NAME=$1; shift
AB="python3 bench.py --no-cpu-baseline --no-host-leg --no-two-streams-leg --no-configs --match-rows 0 --steps 10"
i=0
for v in "$@"; do
    i=$((i + 1))
    if [ -z "$v" ]; then unset "$NAME"; else export "$NAME=$v"; fi
    timeout -k 10 300 $AB > gpurun_out/exp/env_$i.json 2> gpurun_out/exp/env_$i.err || { tail -3 gpurun_out/exp/env_$i.err; exit 1; }
    python3 -c "import json; d = json.load(open('gpurun_out/exp/env_$i.json')); print('$NAME=$v', round(d['value'], 1), d['verified'], d['ms_per_step'])"
done
