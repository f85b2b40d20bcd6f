set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1prof
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --brute 262144 > $O/r1_bench_c2.json 2> $O/bench_c2.err
timeout -k 10 200 python bench.py --no-cpu-baseline --workload c1 > $O/r1_bench_c1.json 2> $O/bench_c1.err
timeout -k 10 200 python bench.py --no-cpu-baseline --workload c0 > $O/r1_bench_c0.json 2> $O/bench_c0.err
timeout -k 10 300 python bench.py --no-cpu-baseline --workload c3 --steps 5 --warmup 2 > $O/r1_bench_c3_single_gpu.json 2> $O/bench_c3.err
timeout -k 10 200 python tools/time_walk_slots.py > $O/r1_walk_alone_by_slots.txt 2> $O/tw.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c2 -- python3 $R/bench.py --no-cpu-baseline > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/tl -o tl -- python3 $R/bench.py --no-cpu-baseline --steps 5 > $O/tl.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/hb_t -o t -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 --no-overlap > $O/hb_t.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format rocpd -d $O/hb_f -o f -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 --no-overlap > $O/hb_f.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format rocpd -d $O/hb_w -o w -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 --no-overlap > $O/hb_w.log 2>&1
cd $R
db() { ls $O/$1/*/*.db $O/$1/*.db 2>/dev/null | head -1; }
python3 tools/timeline.py $(db tl) > $O/r1_step_timeline_c2.txt
python3 tools/timeline.py $(db hb_t) > $O/r1_step_timeline_c2_one_stream.txt
python3 tools/hbm_table.py $(db hb_t) $(db hb_f) $(db hb_w) > $O/r1_hbm_kernels_c2.txt
cp $(ls $O/stats/*/*kernel_stats.csv $O/stats/*kernel_stats.csv 2>/dev/null | head -1) $O/r1_bench_c2_kernel_stats.csv
rm -rf $O/stats $O/tl $O/hb_t $O/hb_f $O/hb_w
ls -la $O
