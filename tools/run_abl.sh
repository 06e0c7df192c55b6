mkdir -p gpurun_out/r04
for f in 0 64 128 192 256 448 512 1024 1536 1984; do
  T2V_LIB=tools/libt2v_ablation.so T2V_DEBUG_FLAGS=$f timeout -k 10 120 python tools/ablate_strip3.py >> gpurun_out/r04/ablate1.txt 2>&1 || exit 1
done
T2V_LIB=tools/libt2v_stamps.so timeout -k 10 200 python tools/stamps.py > gpurun_out/r04/stamps2.txt 2>&1
tail -5 gpurun_out/r04/ablate1.txt
