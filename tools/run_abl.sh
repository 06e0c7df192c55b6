mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/ablate2.txt
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv" > gpurun_out/r04/ops_conv.txt 2>&1; tail -3 gpurun_out/r04/ops_conv.txt
for ks in 0 1; do
for f in 0 64 128 192 256 448 512 1024 1536 1984; do
  if [ $ks = 1 ]; then export T2V_NO_KS=1; else unset T2V_NO_KS; fi
  echo "NO_KS=$ks" >> gpurun_out/r04/ablate2.txt
  T2V_LIB=tools/libt2v_ablation.so T2V_DEBUG_FLAGS=$f timeout -k 10 120 python tools/ablate_strip3.py >> gpurun_out/r04/ablate2.txt 2>&1 || exit 1
done
done
unset T2V_NO_KS
T2V_LIB=tools/libt2v_stamps.so timeout -k 10 200 python tools/stamps.py > gpurun_out/r04/stamps2.txt 2>&1
T2V_NO_KS=1 T2V_LIB=tools/libt2v_stamps.so timeout -k 10 200 python tools/stamps.py > gpurun_out/r04/stamps2_noks.txt 2>&1
grep flags gpurun_out/r04/ablate2.txt | tail -5
