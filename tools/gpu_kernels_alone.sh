# usage: gpu_kernels_alone.sh OUTDIR [workload] : one step's kernel timeline with every stage on one stream
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
W=${2:-c2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/tla -o tl -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 3 --warmup 2 --no-overlap --workload $W > $O/tla.log 2>&1 || { tail $O/tla.log; exit 1; }
cd $R
python3 tools/timeline.py $(ls $O/tla/*/*.db $O/tla/*.db 2>/dev/null | head -1) > $O/timeline_${W}_one_stream.txt
rm -rf $O/tla
