# usage: gpu_check.sh OUTDIR : parity tests, the default bench line (no CPU baseline), C3, and one step's kernel timeline
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 > $O/c2.json 2> $O/c2.err || { tail $O/c2.err; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 --workload c3 --steps 5 --warmup 2 > $O/c3.json 2> $O/c3.err || { tail $O/c3.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format rocpd -d $O/tl -o tl -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --brute 0 --steps 5 > $O/tl.log 2>&1 || { tail $O/tl.log; exit 1; }
cd $R
python3 tools/timeline.py $(ls $O/tl/*/*.db $O/tl/*.db 2>/dev/null | head -1) > $O/timeline_c2.txt
rm -rf $O/tl
python3 - "$O" <<'PY'
import json, sys
for f in ("c2", "c3"):
    d = json.load(open(sys.argv[1] + "/%s.json" % f)); r = d["roofline"]
    print(f, round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["stage_ms"].items()}, "walk frac", round(r["frac"], 4),
          "alone", round((r.get("device_to_itself") or {}).get("kernel_ms", 0), 3))
PY
