# Kernel trace of the replayed training step at 64 and 8 clips: per-step launch summaries under gpurun_out/qt/.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qt; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --single-dtype --no-extra > $O/bench_traced.json 2> $O/kt.log || exit 1
python3 $R/tools/step_trace.py $(ls $O/kt/*/*kernel_trace.csv | head -1) 5 sequence > $O/step_launch_summary.txt 2>&1
rm -rf $O/kt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt8 -- python3 $R/bench.py --batch 8 --no-cpu-baseline --single-dtype --no-extra > $O/bench_batch8_traced.json 2> $O/kt8.log || exit 1
python3 $R/tools/step_trace.py $(ls $O/kt8/*/*kernel_trace.csv | head -1) 5 > $O/step_launch_summary_batch8.txt 2>&1
rm -rf $O/kt8
echo QT_DONE
