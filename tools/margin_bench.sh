# usage: margin_bench.sh OUTDIR workload [margins...] : bench line of a workload with the contact solver's
# first-hypothesis margin forced (-1 = the library's own switch)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
W=$2
shift; shift
M=${@:-"-1 0.5 1 2"}
mkdir -p $O
cd $R
for m in $M; do
  PSIM_DETECT_MARGIN=$m timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 --no-walk-alone --workload $W --steps 10 --warmup 4 > $O/${W}_m$m.json 2> $O/${W}_m$m.err || { tail -5 $O/${W}_m$m.err; exit 1; }
done
python3 - "$O" "$W" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/" + sys.argv[2] + "_m*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-22s %7.3f ms/step  %s  %s" % (f.split("/")[-1], d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()}, d["last_step"]))
PY
