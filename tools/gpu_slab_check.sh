# usage: gpu_slab_check.sh OUTDIR : the band-split collision stage on the GPU (tests), the default bench line
# without the CPU leg, and one rank's share of C3 in an 8-rank job
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_slab.py -x -v --timeout 300 --timeout-method thread > $O/slab.log 2>&1 || { tail -40 $O/slab.log; exit 1; }
tail -3 $O/slab.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 > $O/c2.json 2> $O/c2.err || { tail $O/c2.err; exit 1; }
for r in 0 3; do
timeout -k 10 300 python bench.py --emulate-rank $r --world 8 --steps 5 --warmup 2 > $O/c3_rank${r}_of_8.json 2> $O/c3_rank$r.err || { tail $O/c3_rank$r.err; exit 1; }
done
timeout -k 10 300 python bench.py --staged --workload c3 --no-cpu-baseline --no-other-workloads --brute 0 --steps 5 --warmup 2 > $O/c3_staged_one_rank.json 2> $O/c3_staged.err || { tail $O/c3_staged.err; exit 1; }
python3 - "$O" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/c2.json")); print("c2", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["stage_ms"].items()})
for r in (0, 3):
    d = json.load(open(sys.argv[1] + "/c3_rank%d_of_8.json" % r)); e = d["emulated_rank"]
    print("c3 rank", r, round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in e["ms_per_step_by_piece"].items()}, e["exchange_bytes"], e["rows"], "fallbacks", e["fallbacks"])
d = json.load(open(sys.argv[1] + "/c3_staged_one_rank.json")); print("c3 staged one rank", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["stage_ms"].items()})
PY
