mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > gpurun_out/r04/t4.txt 2>&1; tail -3 gpurun_out/r04/t4.txt
T2V_LIB=tools/libt2v_stamps.so timeout -k 10 200 python tools/stamps.py > gpurun_out/r04/stamps4.txt 2>&1
python3 bench.py --no_cpu_baseline --no_extra --no_hbm > gpurun_out/r04/bench4.log 2>&1; tail -c 300 gpurun_out/r04/bench4.log
