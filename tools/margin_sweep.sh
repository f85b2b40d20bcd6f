# usage: margin_sweep.sh OUTDIR [margins...] : the contact solver's first-hypothesis margin (-1 = the library's own
# switch) on the dense and clumped cases
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
M=${@:-"-1 0"}
mkdir -p $O
cd $R
for m in $M; do
  echo "== margin $m" >> $O/long.txt
  PSIM_DETECT_MARGIN=$m timeout -k 10 200 python tools/long_run.py 600 >> $O/long.txt 2>&1 || { tail -5 $O/long.txt; exit 1; }
done
grep -E "margin|step +(100|300|600)" $O/long.txt
for m in $M; do
  echo "== margin $m" >> $O/dense.txt
  for n in 160000 320000; do
    PSIM_DETECT_MARGIN=$m timeout -k 10 120 python tools/dense_blob.py $n >> $O/dense.txt 2>&1 || { tail -5 $O/dense.txt; exit 1; }
  done
done
grep -E "margin|step 2" $O/dense.txt
