# kernel trace of the 1 M x 1 M search (bench_match.py) with the in-tree library and, when given, with
# gpurun_exp/<name>.so:  gpurun -- 'bash tools/trace_match.sh [name]'
set -o pipefail
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/rank
mkdir -p $O
for v in default "$@"; do
  if [ "$v" = default ]; then L=; else L=$R/gpurun_exp/$v.so; fi
  export MIF_LIBRARY=$L
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$v -o t -- python $R/tools/bench_match.py --nq 1000000 --nd 1000000 --steps 2 --warmup 1 > $O/t_$v.log 2>&1 || exit 1
  echo "== $v"; python $R/tools/kstats.py $O/t_$v/t_kernel_stats.csv 3 8 | cut -c1-140
done
