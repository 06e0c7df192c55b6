mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "bmm or nonlocal or attention or multi" > gpurun_out/r04/t1.txt 2>&1; tail -3 gpurun_out/r04/t1.txt
timeout -k 10 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "attention3d or resnet3d or down_block or train_steps_uncond or benchmark_iteration_B32_vs_oracle or pooled_and_unpooled" > gpurun_out/r04/t2.txt 2>&1; tail -5 gpurun_out/r04/t2.txt
python3 tools/d_roofline.py > gpurun_out/r04/d_roofline2.log 2>&1; tail -c 300 gpurun_out/r04/d_roofline2.log
T2V_NO_SKIP_POOLS_FIRST=1 python3 tools/d_roofline.py > gpurun_out/r04/d_roofline2_noskip.log 2>&1; tail -c 300 gpurun_out/r04/d_roofline2_noskip.log
python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm > gpurun_out/r04/bench2.log 2>&1; tail -c 300 gpurun_out/r04/bench2.log
T2V_NO_SKIP_POOLS_FIRST=1 python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > gpurun_out/r04/bench2_noskip.log 2>&1; tail -c 300 gpurun_out/r04/bench2_noskip.log
