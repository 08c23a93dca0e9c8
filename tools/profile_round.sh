set -x
# (the *_clock.py tools below load diagnostic builds from tools/_build/: run tools/build_stamp_libs.sh in the container first, after any change to csrc/)
R=$PWD; O=$R/gpurun_out/${ROUND_DIR:-r6fin}; mkdir -p $O
python bench.py --steps 200 > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --steps 100 --no_hip_graph --no_cpu_baseline --no_extra_sizes > $O/bench_eager.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
# kernel durations in stream order (what bench.py's per-launch events measure; the default at this size since round 3)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 20 --warmup 2 --no_cpu_baseline --no_extra_sizes --no_exact_split --no_hip_graph > $O/bench_under_rocprof.json 2>/dev/null
rm -f $O/stats/*kernel_trace.csv
B="python3 $R/bench.py --steps 4 --warmup 1 --no_prefill --no_cpu_baseline --no_hip_graph --no_extra_sizes --no_exact_split --no_repeats"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/pmc_a -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_b -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_c -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_d -o p -- $B > /dev/null 2>&1
# BASELINE config 5: memory-side traffic of the L-BFGS sweeps at FULL history (the run fills it first; the last 4 launches count)
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/pmc_nin -o p -- python3 $R/bench.py --model nin --steps 4 --warmup 1 --no_cpu_baseline --no_hip_graph --no_extra_sizes --no_exact_split --no_repeats > /dev/null 2>&1
cd $R
python tools/pmc_summary.py --last 4 $O/pmc_traffic_nin.json $O/pmc_nin > $O/pmc_summary_nin.txt 2>&1
rm -rf $O/pmc_nin
python tools/pmc_summary.py $O/pmc_traffic.json $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d > $O/pmc_summary.txt 2>&1
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d
for S in 256 512 724 1024 1448 2048; do python bench.py --size $S --steps 100 --no_cpu_baseline --no_extra_sizes 2>/dev/null >> $O/sizes_lbfgs.jsonl; done
for S in 1024 1448 2048; do python bench.py --size $S --optimizer adam --steps 100 --no_cpu_baseline --no_extra_sizes 2>/dev/null >> $O/sizes_adam.jsonl; done
python tools/run_configs.py --configs 2,3,4,5,6 --out $O/configs.json > $O/configs.log 2>&1
# per-dispatch durations (and idle gaps) of one steady-state iteration as the product runs it - replayed from the hipGraph, history full:
# the per-launch tables (eager events overstate host-bound launches)
for S in 1024 512 256; do bash tools/trace_gaps.sh gpurun_out/${ROUND_DIR:-r6fin}/tg$S --size $S --steps 130 > $O/launches_graph_$S.txt 2>&1; done
bash tools/trace_gaps.sh gpurun_out/${ROUND_DIR:-r6fin}/tgnin --model nin --steps 130 > $O/launches_graph_nin.txt 2>&1
python tools/graph_host_cost.py 256 512 1024 2>/dev/null | grep size > $O/graph_host_cost.txt
python tools/check_x3p.py 1024 5 10 > $O/check_x3p.txt 2>&1
python tools/x3p_clock.py 64 64 1024 plain > $O/clock_x3p_conv1_2.txt 2>&1
python tools/x3p_clock.py 64 64 1024 masked >> $O/clock_x3p_conv1_2.txt 2>&1
python tools/x3p_clock.py 256 256 256 plain >> $O/clock_x3p_conv1_2.txt 2>&1
python tools/soak_kernels.py 3000 > $O/soak.txt 2>&1
(tools/mfma_probe/stage_bw 64 1024; tools/mfma_probe/stage_bw 128 512) > $O/stage_bw.txt 2>&1
# the reproducer of round 4's Gram fault in every build of probes_r05.md section 4 (tools/mfma_probe/build_zero_lanes.sh in the container first)
(cd tools/mfma_probe; for b in n0 p4 p1 p2 p3 il g2 pad g2s pads n0s g2p padp g2pc padpc g2ma padma one g2one; do echo "== $b"; timeout 300 ./gram128_zero_lanes_$b 512 16384 2000 600 | tail -1; done; ./mfma_src_war) > $O/gram128_zero_lanes.txt 2>&1
python tools/x3w_clock.py 512 512 128 > $O/clock_conv4_2.txt 2>&1
python tools/x3w_clock.py 64 64 1024 > $O/clock_conv1_2.txt 2>&1
python tools/bench_x3w.py 1024 5 10 > $O/x3_vs_x3w.txt 2>&1
python tools/x3q_clock.py 512 512 128 > $O/clock_x3q_conv4_2.txt 2>&1
python tools/bench_x3q.py 1024 5 10 > $O/x3w_vs_x3q.txt 2>&1
python tools/stress_x3q.py 30 > $O/stress_x3q.txt 2>&1
python bench.py --model nin --steps 200 --no_cpu_baseline > $O/bench_nin.json 2>/dev/null
python tools/lbfgs_clock.py 196608 100 > $O/clock_lbfgs.txt 2>&1
python tools/bench_fused_gram.py 1024 20 > $O/fused_gram.txt 2>&1
python tools/stress_fused.py 100 > $O/stress_fused.txt 2>&1
python tools/bench_gram.py > $O/bench_gram.txt 2>/dev/null
python tools/stress_gram.py 100 > $O/stress_gram.txt 2>&1
ls -la $O; tail -c 600 $O/bench_graph.json
