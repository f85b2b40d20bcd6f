# usage: trace_cmd.sh OUTDIR python-script [args...] : kernel totals and timeline of the last full step the script runs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/tl -o tl -- python3 "$@" > $O/run.txt 2> $O/run.err || { tail $O/run.err; exit 1; }
cd $R
DB=$(ls $O/tl/*/*.db $O/tl/*.db 2>/dev/null | head -1)
python3 tools/timeline_summary.py $DB > $O/summary.txt
python3 tools/timeline.py $DB > $O/timeline.txt
rm -rf $O/tl
tail -4 $O/run.txt
head -40 $O/summary.txt
