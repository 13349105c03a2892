# Per-frame stage times of the frame pipeline at several batch sizes (one gpurun call):
#   gpurun -- "bash tools/exp_batch.sh"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp
for b in 4 8 16 32 64 512; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-verify --no-host-leg --no-two-streams-leg --no-configs --match-rows 0 --batch $b --steps 20 --warmup 5 > gpurun_out/exp/bb_$b.json 2>/dev/null || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/exp/bb_$b.json')); s=d['stage_ms_per_batch']; print($b, round(d['value'],1), {k: round(v/$b*1000,2) for k,v in s.items()}, 'us per frame')"
done
