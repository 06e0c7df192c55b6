R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "attention or resnet3d or train_steps or first_step_grad or two_rank_replicas or teacher" > $O/t12.txt 2>&1; tail -3 $O/t12.txt
timeout -k 10 300 python tools/accum_sites.py 32 > $O/accum_sites12.txt 2>&1; tail -4 $O/accum_sites12.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_p -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_d_roofline --no_extra --no_roofline --no_hbm > $O/trace_p.log 2>&1
grep -i "at::native\|rocclr" $O/trace_p/*/*_kernel_stats.csv | cut -c1-200
ms=$(grep -o '"ms_per_step": [0-9.]*' $O/trace_p.log | head -1 | cut -d' ' -f2); echo "ms under profiler $ms"
python3 $R/tools/gap_analysis.py $O/trace_p $ms | head -3
rm -rf $O/trace_p
