R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_models_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "attention or nonlocal or resnet3d or train_steps or double_backward or down_block or gp" > $O/t11.txt 2>&1; tail -3 $O/t11.txt
timeout -k 10 300 python tools/accum_sites.py 32 > $O/accum_sites11.txt 2>&1; tail -30 $O/accum_sites11.txt
python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > $O/bench11.log 2>&1; tail -c 200 $O/bench11.log
