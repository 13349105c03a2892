#!/bin/bash
# Kernel trace and two counter passes of the SURF batch with the one-launch integral image forced
# (MIF_SURF_INTEGRAL=fused), printing the lines of k_integral_fused:
#   gpurun -- 'bash tools/exp_fused_prof.sh'      (writes under gpurun_out/fused_*)
export TMPDIR=/tmp; R=$PWD; cd /tmp
export MIF_SURF_INTEGRAL=fused
SURF="python3 $R/tools/bench_surf.py --skip-single --batch 64 --steps 5"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fused_trace -o t -- $SURF > $R/gpurun_out/fused_trace.log 2>&1 || exit 1
python3 $R/tools/kstats.py $R/gpurun_out/fused_trace/t_kernel_stats.csv 7 3
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/fused_pmc -o q -- $SURF > $R/gpurun_out/fused_pmc.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $R/gpurun_out/fused_pmc/q_counter_collection.csv | grep "k_integral"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAVES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/fused_pmc2 -o q -- $SURF > $R/gpurun_out/fused_pmc2.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $R/gpurun_out/fused_pmc2/q_counter_collection.csv | grep "k_integral"
