#!/bin/bash
# Regenerates the round-4 evidence under gpurun_out/r04 (copied into profiles/ afterwards). Run on the GPU box from the repo root:
#   bash tools/r04_profiles.sh [quick]      (quick: the kernel trace + timeline of the default benchmark only)
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04
mkdir -p $O; rm -rf $O/trace $O/pmc_* $O/trace_*
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_d_roofline --no_extra > $O/trace.log 2>&1 && echo "trace ok"
rc=$?
cp $O/trace/*/*_kernel_stats.csv $O/r04_kernel_stats.csv 2>/dev/null
python3 $R/tools/prof_summary.py $O/trace > $O/r04_kernel_trace_summary.txt 2>&1
ms=$(grep -o '"ms_per_step": [0-9.]*' $O/trace.log | head -1 | cut -d' ' -f2)
python3 $R/tools/gap_analysis.py $O/trace $ms > $O/r04_replay_timeline.txt 2>&1
python3 $R/tools/phase_times.py $O/trace $ms > $O/r04_phase_times.txt 2>&1
rm -rf $O/trace
[ "$1" = "quick" ] && exit $rc
cd $R
# PMC passes first (counters in their own runs, kernel trace only): the default line below quotes the traffic file (same conv.hip: same sha1)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/conv_micro.py both 3 > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/conv_micro.py both 3 > $O/pmc_write.log 2>&1 &&
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/r04_pmc_traffic.json && echo "traffic ok"
cp $O/r04_pmc_traffic.json profiles/r04_pmc_traffic.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_hbm_fetch -- python3 tools/hbm_micro.py 3 > $O/pmc_hbm_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_hbm_write -- python3 tools/hbm_micro.py 3 > $O/pmc_hbm_write.log 2>&1 &&
python3 tools/pmc_hbm.py $O/pmc_hbm_fetch $O/pmc_hbm_write > $O/r04_pmc_hbm.json && echo "hbm pmc ok"
cp $O/r04_pmc_hbm.json profiles/r04_pmc_hbm.json
# per-launch shape table of one iteration's convolution launches (T2V_PROF_DUMP) + the default line
T2V_PROF_DUMP=$O/r04_conv_launches.csv python3 bench.py > $O/bench_default.log 2>&1 && tail -1 $O/bench_default.log > $O/r04_bench_default.json && echo "bench ok"
python3 tools/launch_table.py $O/r04_conv_launches.csv 5 > $O/r04_conv_launch_shapes.txt 2>&1
# D forward+backward (the north star's own line) and BASELINE configs[2] under the profiler
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_d -- python3 $R/tools/d_roofline.py > $O/d_roofline.log 2>&1 && echo "d ok"
cp $O/trace_d/*/*_kernel_stats.csv $O/r04_d_fwdbwd_kernel_stats.csv 2>/dev/null; rm -rf $O/trace_d
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python3 $R/bench.py --cond --bf16 --steps 10 --warmup 3 --no_cpu_baseline --no_roofline --no_d_roofline --no_extra > $O/cfg2.log 2>&1 && echo "cfg2 ok"
cp $O/trace_c2/*/*_kernel_stats.csv $O/r04_cfg2_kernel_stats.csv 2>/dev/null
python3 $R/tools/gap_analysis.py $O/trace_c2 $(grep -o '"ms_per_step": [0-9.]*' $O/cfg2.log | head -1 | cut -d' ' -f2) > $O/r04_cfg2_replay_timeline.txt 2>&1
rm -rf $O/trace_c2
# SQ counters of the dominant kernels (their own pass)
cd $R
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq -- python3 tools/conv_micro.py both 3 > $O/pmc_sq.log 2>&1 &&
python3 tools/pmc_sq.py $O/pmc_sq > $O/r04_pmc_sq.json && echo "sq ok"
python3 tools/conv_micro.py both 20 > $O/r04_conv_micro.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_hbm_fetch $O/pmc_hbm_write
exit $rc
