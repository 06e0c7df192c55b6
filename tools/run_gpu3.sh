mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > gpurun_out/r04/t3.txt 2>&1; tail -3 gpurun_out/r04/t3.txt
python3 tools/d_roofline.py > gpurun_out/r04/d_roofline3.log 2>&1; tail -c 200 gpurun_out/r04/d_roofline3.log
T2V_NO_WGRAD_THIN=1 python3 tools/d_roofline.py > gpurun_out/r04/d_roofline3_nothin.log 2>&1; tail -c 200 gpurun_out/r04/d_roofline3_nothin.log
python3 tools/conv_suite.py 10 "stem conv1" > gpurun_out/r04/suite3.txt 2>&1; cat gpurun_out/r04/suite3.txt
T2V_NO_WGRAD_THIN=1 python3 tools/conv_suite.py 10 "stem conv1" > gpurun_out/r04/suite3_nothin.txt 2>&1; cat gpurun_out/r04/suite3_nothin.txt
python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > gpurun_out/r04/bench3.log 2>&1; tail -c 200 gpurun_out/r04/bench3.log
