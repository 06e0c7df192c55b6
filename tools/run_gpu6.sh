mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "benchmark_iteration" > gpurun_out/r04_t6.txt 2>&1; tail -3 gpurun_out/r04_t6.txt
T2V_LIB=tools/libt2v_stamps.so timeout -k 10 200 python tools/stamps.py deep_d3 deep_d2 deep_d2c1 > gpurun_out/r04_stamps6.txt 2>&1
bash tools/r04_profiles.sh quick > gpurun_out/r04_quick.log 2>&1
head -70 gpurun_out/r04/r04_replay_timeline.txt > gpurun_out/r04_timeline_head.txt
tail -5 gpurun_out/r04_quick.log
