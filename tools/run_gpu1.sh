mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "bmm or nonlocal or attention or multi or bf16_exchange or arena" > gpurun_out/r04/t1.txt 2>&1; tail -3 gpurun_out/r04/t1.txt
timeout -k 10 300 python -m pytest tests/test_models_gpu.py -x -q -m gpu -k "attention3d or resnet3d" > gpurun_out/r04/t2.txt 2>&1; tail -3 gpurun_out/r04/t2.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04/trace_d -- python3 $GRAFT_REPO_ROOT/tools/d_roofline.py > $GRAFT_REPO_ROOT/gpurun_out/r04/d_roofline1.log 2>&1
cd $GRAFT_REPO_ROOT
cp gpurun_out/r04/trace_d/*/*_kernel_stats.csv gpurun_out/r04/d_kernel_stats1.csv 2>/dev/null; rm -rf gpurun_out/r04/trace_d
python3 tools/d_roofline.py > gpurun_out/r04/d_roofline1_noprof.log 2>&1
tail -c 400 gpurun_out/r04/d_roofline1_noprof.log
