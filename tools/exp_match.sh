#!/bin/bash
# A/B of library builds on the 1 M x 1 M search (tools/bench_match.py) inside ONE gpurun call:
#   gpurun -- 'bash tools/exp_match.sh default cand default cand'   (names as in tools/exp_run.sh)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
R=$PWD
O=$R/gpurun_out/exp
mkdir -p "$O"
i=0
for v in "$@"; do
    i=$((i + 1))
    if [ "$v" = default ]; then L=; else L=$R/gpurun_exp/$v.so; fi
    MIF_LIBRARY=$L timeout -k 10 300 python3 $R/tools/bench_match.py --nq 1000000 --nd 1000000 --steps 3 --warmup 1 > "$O/m_${i}_$v.json" 2> "$O/m_${i}_$v.err" || exit 1
    python3 -c "import json; d = json.load(open('$O/m_${i}_$v.json')); print('$v', round(d['value']), {k: d[k] for k in d if k not in ('metric', 'unit', 'config', 'roofline', 'value')})"
done
