R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "gen or up_block or render or train_steps or batch_norm or bn or graph_replay" > $O/t10.txt 2>&1; tail -3 $O/t10.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_p -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_d_roofline --no_extra --no_roofline --no_hbm > $O/trace_p.log 2>&1
ms=$(grep -o '"ms_per_step": [0-9.]*' $O/trace_p.log | head -1 | cut -d' ' -f2)
python3 $R/tools/phase_times.py $O/trace_p $ms > $O/r04_phase_times.txt 2>&1; cat $O/r04_phase_times.txt | cut -c1-250
python3 $R/tools/gap_analysis.py $O/trace_p $ms > $O/timeline10.txt 2>&1
rm -rf $O/trace_p
cd $R
python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > $O/bench10.log 2>&1; tail -c 200 $O/bench10.log
