# The measurement set kept under profiles/ (written under gpurun_out/r3prof and copied from there).
# usage (on the GPU box): bash tools/collect_profiles.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3prof
P=r3
mkdir -p $O
cd $R
step() { echo "== $*" >&2; }
step bench lines
timeout -k 10 400 python bench.py > $O/${P}_bench_c2.json 2> $O/bench_c2.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 --workload c1 > $O/${P}_bench_c1.json 2> $O/bench_c1.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 --workload c0 > $O/${P}_bench_c0.json 2> $O/bench_c0.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 --workload c3 --steps 5 --warmup 2 > $O/${P}_bench_c3_single_gpu.json 2> $O/bench_c3.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 --workload c3 --steps 5 --warmup 2 --staged > $O/${P}_bench_c3_staged_one_rank.json 2> $O/bench_c3s.err || exit 1
step one rank of eight
for r in 0 3 7; do
timeout -k 10 300 python bench.py --emulate-rank $r --world 8 --steps 5 --warmup 2 > $O/${P}_bench_c3_rank${r}_of_8_emulated.json 2> $O/emu$r.err || exit 1
done
timeout -k 10 300 python bench.py --emulate-rank 1 --world 2 --steps 5 --warmup 2 > $O/${P}_bench_c3_rank1_of_2_emulated.json 2> $O/emu2.err || exit 1
timeout -k 10 300 python bench.py --emulate-rank 1 --world 4 --steps 5 --warmup 2 > $O/${P}_bench_c3_rank1_of_4_emulated.json 2> $O/emu4.err || exit 1
step walk lane profile
timeout -k 10 200 python tools/walk_profile.py c2 > $O/${P}_walk_lane_profile_c2.json 2> $O/wp.err || exit 1
step walk alone by slots
timeout -k 10 200 python tools/time_walk_slots.py > $O/${P}_walk_alone_by_slots.txt 2> $O/tw.err || exit 1
step stress runs
timeout -k 10 300 python tools/long_run.py 600 > $O/${P}_long_run_c2_600_steps.txt 2> $O/lr.err || exit 1
timeout -k 10 300 python tools/big_smoke.py 67108864 3 c4 > $O/${P}_smoke_c4_64Mi_single_gpu.txt 2> $O/c4.err || exit 1
timeout -k 10 300 python tools/big_smoke.py 67108864 2 > $O/${P}_smoke_64Mi_single_gpu.txt 2> $O/s64.err || exit 1
timeout -k 10 200 python tools/drift_report.py > $O/${P}_drift_c0.txt 2> $O/drift.err || exit 1
cd /tmp && export TMPDIR=/tmp
step kernel stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c2 -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --no-walk-alone --brute 0 > $O/stats.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -o c3 -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --no-walk-alone --brute 0 --workload c3 --steps 5 --warmup 2 > $O/stats3.log 2>&1 || exit 1
step timelines
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/tl -o tl -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 5 > $O/tl.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/tl3 -o tl -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 3 --warmup 2 --workload c3 > $O/tl3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/hb_t -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 3 --warmup 2 --no-overlap > $O/hb_t.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/hb3_t -o t -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 3 --warmup 2 --no-overlap --workload c3 > $O/hb3_t.log 2>&1 || exit 1
step counters
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format rocpd -d $O/hb_f -o f -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 3 --warmup 2 --no-overlap > $O/hb_f.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format rocpd -d $O/hb_w -o w -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 3 --warmup 2 --no-overlap > $O/hb_w.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format rocpd -d $O/pmc -o p -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 2 --warmup 1 --no-overlap > $O/pmc.log 2>&1 || exit 1
cd $R
db() { ls $O/$1/*/*.db $O/$1/*.db 2>/dev/null | head -1; }
python3 tools/timeline.py $(db tl) > $O/${P}_step_timeline_c2.txt
python3 tools/timeline.py $(db tl3) > $O/${P}_step_timeline_c3.txt
python3 tools/timeline.py $(db hb_t) > $O/${P}_step_timeline_c2_one_stream.txt
python3 tools/timeline.py $(db hb3_t) > $O/${P}_step_timeline_c3_one_stream.txt
python3 tools/hbm_table.py $(db hb_t) $(db hb_f) $(db hb_w) > $O/${P}_hbm_kernels_c2.txt
python3 tools/hbm_table.py $(db hb3_t) --n 16777216 > $O/${P}_hbm_kernels_c3.txt
python3 tools/pmc_summary.py $(db pmc) gravity_walk_multi > $O/${P}_walk_pmc_raw.txt
python3 tools/pmc_summary.py $(db hb_f) gravity_walk_multi > $O/${P}_walk_fetch_raw.txt
python3 tools/pmc_summary.py $(db hb_w) gravity_walk_multi > $O/${P}_walk_write_raw.txt
cp $(ls $O/stats/*/*kernel_stats.csv $O/stats/*kernel_stats.csv 2>/dev/null | head -1) $O/${P}_bench_c2_kernel_stats.csv
cp $(ls $O/stats3/*/*kernel_stats.csv $O/stats3/*kernel_stats.csv 2>/dev/null | head -1) $O/${P}_bench_c3_kernel_stats.csv
rm -rf $O/stats $O/stats3 $O/tl $O/tl3 $O/hb_t $O/hb3_t $O/hb_f $O/hb_w $O/pmc
ls -la $O
