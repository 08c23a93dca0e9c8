set -x
# The part of tools/profile_round.sh that the bench line, the roofline and the per-launch tables come from (no soaks, clocks or probe matrices):
# what has to be refreshed when a kernel's default route changes late in a round.  tools/collect_profiles.py picks the files up.
R=$PWD; O=$R/gpurun_out/${ROUND_DIR:-r6fin}; mkdir -p $O
python bench.py --steps 200 > $O/bench_graph.json 2> $O/bench_graph.err
python bench.py --steps 100 --no_hip_graph --no_cpu_baseline --no_extra_sizes > $O/bench_eager.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 20 --warmup 2 --no_cpu_baseline --no_extra_sizes --no_exact_split --no_hip_graph > $O/bench_under_rocprof.json 2>/dev/null
rm -f $O/stats/*kernel_trace.csv
B="python3 $R/bench.py --steps 4 --warmup 1 --no_prefill --no_cpu_baseline --no_hip_graph --no_extra_sizes --no_exact_split --no_repeats"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/pmc_a -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_b -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_c -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_d -o p -- $B > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/pmc_nin -o p -- python3 $R/bench.py --model nin --steps 4 --warmup 1 --no_cpu_baseline --no_hip_graph --no_extra_sizes --no_exact_split --no_repeats > /dev/null 2>&1
cd $R
python tools/pmc_summary.py --last 4 $O/pmc_traffic_nin.json $O/pmc_nin > $O/pmc_summary_nin.txt 2>&1
rm -rf $O/pmc_nin
python tools/pmc_summary.py $O/pmc_traffic.json $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d > $O/pmc_summary.txt 2>&1
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d
rm -f $O/sizes_lbfgs.jsonl $O/sizes_adam.jsonl
for S in 256 512 724 1024 1448 2048; do python bench.py --size $S --steps 100 --no_cpu_baseline --no_extra_sizes 2>/dev/null >> $O/sizes_lbfgs.jsonl; done
for S in 1024 1448 2048; do python bench.py --size $S --optimizer adam --steps 100 --no_cpu_baseline --no_extra_sizes 2>/dev/null >> $O/sizes_adam.jsonl; done
python tools/run_configs.py --configs 2,3,4,5,6 --out $O/configs.json > $O/configs.log 2>&1
for S in 1024 512 256; do bash tools/trace_gaps.sh gpurun_out/${ROUND_DIR:-r6fin}/tg$S --size $S --steps 130 > $O/launches_graph_$S.txt 2>&1; done
bash tools/trace_gaps.sh gpurun_out/${ROUND_DIR:-r6fin}/tgnin --model nin --steps 130 > $O/launches_graph_nin.txt 2>&1
python tools/graph_host_cost.py 256 512 1024 2>/dev/null | grep size > $O/graph_host_cost.txt
python tools/check_x3p.py 1024 5 10 > $O/check_x3p.txt 2>&1
python bench.py --model nin --steps 200 --no_cpu_baseline > $O/bench_nin.json 2>/dev/null
ls -la $O; tail -c 600 $O/bench_graph.json
