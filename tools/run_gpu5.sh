mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04/t5_full.txt 2>&1; tail -5 gpurun_out/r04/t5_full.txt
timeout -k 10 300 python tools/accum_sites.py 32 > gpurun_out/r04/accum_sites.txt 2>&1; tail -40 gpurun_out/r04/accum_sites.txt
