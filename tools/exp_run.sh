# experiment runner for one gpurun call (edited per experiment; the kept evidence goes to profiles/)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/exp4
mkdir -p $O
AB="python3 $R/bench.py --no-cpu-baseline --no-verify --no-host-leg --no-two-streams-leg --no-configs --match-rows 0 --steps 10"
for v in default desc_tm default desc_tm; do
  if [ $v = default ]; then timeout -k 10 300 $AB > $O/ab_$v.json 2>/dev/null; else MIF_LIBRARY=$R/gpurun_exp/$v.so timeout -k 10 300 $AB > $O/ab_$v.json 2>/dev/null; fi
  python3 -c "import json; d=json.load(open('$O/ab_$v.json')); print('$v', round(d['value'],1), d['stage_ms_per_batch'])"
done
bash tools/profile_round.sh r03 trace pmc sq surf 4k micro 2>&1 | cut -c1-260
