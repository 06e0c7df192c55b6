R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "pool or up_conv or resnet3d or up_block or train_steps_uncond" > $O/t14.txt 2>&1; tail -3 $O/t14.txt
python3 tools/conv_micro.py pool 20 > $O/micro14.txt 2>&1; grep "box-sum" $O/micro14.txt
T2V_NO_BOXSUM4=1 python3 tools/conv_micro.py pool 20 > $O/micro14_old.txt 2>&1; grep "box-sum" $O/micro14_old.txt
python3 tools/d_roofline.py > $O/d14.log 2>&1; tail -c 250 $O/d14.log
python3 bench.py --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > $O/bench14.log 2>&1; tail -c 150 $O/bench14.log
