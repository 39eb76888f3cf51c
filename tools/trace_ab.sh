# Kernel trace of the replayed training step with and without conv_wide, same box: mean duration per kernel name.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_ab; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in wide nowide; do
  if [ $e = nowide ]; then export STGCN_NO_CONV_WIDE=1; else unset STGCN_NO_CONV_WIDE; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$e -- python3 $R/bench.py --no-cpu-baseline --single-dtype --no-extra > $O/$e.json 2> $O/$e.log || exit 1
  KS=$(ls $O/$e/*/*kernel_stats.csv | head -1); head -14 $KS | cut -c1-150 > $O/${e}_stats.txt
  rm -rf $O/$e
done
echo TRACE_AB_DONE
