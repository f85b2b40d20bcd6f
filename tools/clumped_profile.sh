# usage: clumped_profile.sh OUTDIR [steps] : kernel totals and timeline of the last step of tools/long_run.py
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
N=${2:-600}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format rocpd -d $O/tl -o tl -- python3 $R/tools/long_run.py $N > $O/long_run.txt 2> $O/long_run.err || { tail $O/long_run.err; exit 1; }
cd $R
DB=$(ls $O/tl/*/*.db $O/tl/*.db 2>/dev/null | head -1)
python3 tools/timeline_summary.py $DB > $O/summary.txt
python3 tools/timeline.py $DB > $O/timeline.txt
rm -rf $O/tl
tail -3 $O/long_run.txt
head -45 $O/summary.txt
