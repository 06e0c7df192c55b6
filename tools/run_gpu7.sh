mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv or pool" > gpurun_out/r04/t7.txt 2>&1; tail -3 gpurun_out/r04/t7.txt
timeout -k 10 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "benchmark_iteration_B32_vs_oracle or resnet3d or train_steps_uncond" > gpurun_out/r04/t7b.txt 2>&1; tail -3 gpurun_out/r04/t7b.txt
T2V_LIB=tools/libt2v_stamps.so timeout -k 10 200 python tools/stamps.py deep_d3 deep_d2 > gpurun_out/r04/stamps7.txt 2>&1
python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > gpurun_out/r04/bench7.log 2>&1; tail -c 200 gpurun_out/r04/bench7.log
T2V_NO_MEMBER_SPLITS=1 python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > gpurun_out/r04/bench7_nosplit.log 2>&1; tail -c 200 gpurun_out/r04/bench7_nosplit.log
python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > gpurun_out/r04/bench7b.log 2>&1; tail -c 200 gpurun_out/r04/bench7b.log
