# The commands behind every file of profiles/rNN_*: run on the GPU box as `bash tools/prof_round.sh`, then copy
# gpurun_out/prof/* to profiles/ with the round's prefix.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd $R
# the driver's bench line: bf16 headline + nested fp32 record + both CPU baselines; per-kernel event tables on stderr
timeout -k 10 400 python bench.py --kernel-table > $O/bench.json 2> $O/bench_event_tables.txt || exit 1
# small per-GPU batch (the reference's global 64 over 8 GPUs): graph replay vs launch by launch
timeout -k 10 200 python bench.py --batch 8 --single-dtype --no-cpu-baseline --no-extra > $O/bench_batch8_graph.json 2> /dev/null || exit 1
timeout -k 10 200 python bench.py --batch 8 --single-dtype --no-cpu-baseline --no-graph --no-extra > $O/bench_batch8_nograph.json 2> /dev/null || exit 1
cd /tmp && export TMPDIR=/tmp
# kernel trace of the bench command (graph replay): per-kernel stats, per-step launch summary, the priced kernel's dispatches
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --single-dtype --no-extra > $O/bench_traced.json 2> $O/kt.log || exit 1
KS=$(ls $O/kt/*/*kernel_stats.csv | head -1); cp $KS $O/bench_bf16_kernel_stats.csv
KT=$(ls $O/kt/*/*kernel_trace.csv | head -1); python3 $R/tools/step_trace.py $KT 5 > $O/step_launch_summary.txt 2>&1
python3 $R/tools/step_trace.py $KT 0 > $O/step_launch_summary_launch_by_launch.txt 2>&1
python3 - $KT > $O/bench_bf16_dominant_kernel_dispatches.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_wide_kernel<4, 2, 1, false>" in r["Kernel_Name"]]
print("start_ns,duration_us,grid,workgroup")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    print("%s,%.1f,%s,%s" % (r["Start_Timestamp"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", ""))))
PY
rm -rf $O/kt
# the same at 8 clips per GPU: kernel busy time of a step against its wall time
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt8 -- python3 $R/bench.py --batch 8 --no-cpu-baseline --single-dtype --no-extra > $O/bench_batch8_traced.json 2> $O/kt8.log || exit 1
python3 $R/tools/step_trace.py $(ls $O/kt8/*/*kernel_trace.csv | head -1) 5 > $O/step_launch_summary_batch8.txt 2>&1
rm -rf $O/kt8
# per-kernel counters over a few launch-by-launch steps
timeout -s KILL 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/bench.py --no-cpu-baseline --single-dtype --no-extra --no-graph --steps 3 --warmup 1 > $O/p3.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $(ls $O/p3/*/*counter_collection.csv | head -1) > $O/pmc_per_kernel_summary.txt
rm -rf $O/p3 $O/*.log
# (HBM-side bytes per launch of the step's kernels, FETCH_SIZE / WRITE_SIZE in separate passes: tools/pmc_traffic.sh, a
# GPU call of its own, writes gpurun_out/prof/traffic.json)
echo PROF_DONE
