# usage: sweep_variants.sh OUTDIR [bench args...] : default bench line of every library under variants/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd $R
for f in variants/libpsim_hip_*.so; do
  b=$(basename $f .so)
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --brute 0 --library $R/$f "$@" > $O/$b.json 2> $O/$b.err
done
python3 - "$O" <<'PY'
import json, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print("%-28s %7.3f ms/step  %s  alone %.3f" % (f.split("/")[-1], d["ms_per_step"], {k: round(v, 3) for k, v in d["stage_ms"].items()},
          (r.get("device_to_itself") or {}).get("kernel_ms", 0)))
PY
