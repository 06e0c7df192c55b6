mkdir -p gpurun_out/r04
timeout -k 10 1150 python tools/parity_steps.py 100 gpurun_out/r04/r04_parity_100_steps_cond.json --cond > gpurun_out/r04/parity_cond.log 2>&1; tail -3 gpurun_out/r04/parity_cond.log
