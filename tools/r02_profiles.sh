#!/bin/bash
# Regenerates the round-2 evidence under gpurun_out/ (copied into profiles/ afterwards). Run on the GPU box from the repo root.
set -o pipefail
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_d_roofline --no_extra > $O/trace.log 2>&1 && echo "trace ok" &&
cd $R &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/conv_micro.py both 3 > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/conv_micro.py both 3 > $O/pmc_write.log 2>&1 &&
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/r02_pmc_traffic.json && echo "traffic ok" &&
cp $O/r02_pmc_traffic.json profiles/r02_pmc_traffic.json &&      # the default bench line quotes it (same conv.hip: same sha1)
python3 bench.py > $O/bench_default.log 2>&1 && tail -1 $O/bench_default.log > $O/r02_bench_default.json && echo "bench ok" &&
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq -- python3 tools/conv_micro.py both 3 > $O/pmc_sq.log 2>&1 &&
python3 tools/pmc_sq.py $O/pmc_sq > $O/r02_pmc_sq.json && echo "sq ok" &&
python3 bench.py --size 128 --channels 3 --cond --batch 16 --bf16 --no_cpu_baseline --no_extra --no_d_roofline > $O/cfg4_bf16.log 2>&1 && tail -1 $O/cfg4_bf16.log > $O/r02_bench_cfg4_shape_bf16.json &&
python3 bench.py --size 128 --channels 3 --cond --batch 16 --no_cpu_baseline --no_extra --no_d_roofline > $O/cfg4_f32.log 2>&1 && tail -1 $O/cfg4_f32.log > $O/r02_bench_cfg4_shape_f32.json && echo "cfg4 ok"
rc=$?
cp $O/trace/*/*_kernel_stats.csv $O/r02_kernel_stats.csv 2>/dev/null
python3 tools/prof_summary.py $O/trace > $O/r02_kernel_trace_summary.txt 2>&1
ms=$(grep -o '"ms_per_step": [0-9.]*' $O/trace.log | head -1 | cut -d' ' -f2)
python3 tools/gap_analysis.py $O/trace $ms > $O/r02_replay_timeline.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
# the kernel trace itself is large: keep only the summaries
rm -rf $O/trace
exit $rc
