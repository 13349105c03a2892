# experiment runner for one gpurun call (edited per experiment; the kept evidence goes to profiles/)
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/exp3
mkdir -p $O
trace() { # name, then: env assignments..., --, command...
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  (cd /tmp && env "${envs[@]}" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- "$@" > $O/$name.log 2>&1) || { tail -20 $O/$name.log; return 1; }
  grep "^lib\|frames_per_s\|batch" $O/$name.log | cut -c1-300 | head -5
  python3 tools/kstats.py $O/$name/t_kernel_stats.csv 1 9
}
pmc() { # name, counters (quoted), command...
  name=$1; ctrs=$2; shift 2
  (cd /tmp && timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $O/$name -o q -- "$@" > $O/$name.log 2>&1) || { tail -20 $O/$name.log; return 1; }
  python3 tools/pmc_summary.py $O/$name/q_counter_collection.csv > $O/$name.txt
  grep -v "rocclr\|at::native" $O/$name.txt | cut -c1-400
}
SINGLE="python3 $R/tools/exp_surf_single.py box.pgm"
trace single_new X=1 -- $SINGLE &&
trace single_nokfast MIF_LIBRARY=$R/gpurun_exp/surf_nokfast.so -- $SINGLE &&
trace single_scene X=1 -- python3 $R/tools/exp_surf_single.py scene.pgm &&
trace single_scene_base MIF_LIBRARY=$R/gpurun_exp/base.so -- python3 $R/tools/exp_surf_single.py scene.pgm
