R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_p -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_d_roofline --no_extra --no_roofline --no_hbm > $O/trace_p.log 2>&1
ms=$(grep -o '"ms_per_step": [0-9.]*' $O/trace_p.log | head -1 | cut -d' ' -f2)
python3 $R/tools/phase_times.py $O/trace_p $ms > $O/r04_phase_times.txt 2>&1; cat $O/r04_phase_times.txt
rm -rf $O/trace_p
# BASELINE configs[4], one GPU's share (16x128x128x3, text-conditioned, bf16 compute, per-GPU batch 16): line + kernel stats
cd $R
python3 bench.py --size 128 --channels 3 --cond --bf16 --batch 16 --steps 10 --warmup 3 --no_cpu_baseline --no_d_roofline --no_extra --no_hbm > $O/cfg4.log 2>&1; tail -1 $O/cfg4.log > $O/r04_bench_cfg4_share_bf16.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -- python3 $R/bench.py --size 128 --channels 3 --cond --bf16 --batch 16 --steps 10 --warmup 3 --no_cpu_baseline --no_d_roofline --no_extra --no_hbm --no_roofline > $O/cfg4_prof.log 2>&1
cp $O/trace_c4/*/*_kernel_stats.csv $O/r04_cfg4_share_kernel_stats.csv 2>/dev/null; rm -rf $O/trace_c4
tail -c 300 $O/r04_bench_cfg4_share_bf16.json
