R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "train_steps or graph_replay or uncond_cli or adam_step_counter" > $O/t13.txt 2>&1; tail -3 $O/t13.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_p -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_d_roofline --no_extra --no_roofline --no_hbm > $O/trace_p.log 2>&1
grep -i "at::native" $O/trace_p/*/*_kernel_stats.csv | cut -c1-120
ms=$(grep -o '"ms_per_step": [0-9.]*' $O/trace_p.log | head -1 | cut -d' ' -f2); echo "ms under profiler $ms"
python3 $R/tools/gap_analysis.py $O/trace_p $ms > $O/timeline13.txt 2>&1; head -2 $O/timeline13.txt; grep "at::native" $O/timeline13.txt
rm -rf $O/trace_p
