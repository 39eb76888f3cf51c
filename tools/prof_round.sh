set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --kernel-table > $O/bench_bf16.json 2> $O/bench_bf16_event_table.txt || exit 1
timeout -k 10 300 python bench.py --dtype fp32 --kernel-table > $O/bench_fp32.json 2> $O/bench_fp32_event_table.txt || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/kt.log 2>&1 || exit 1
KS=$(ls $O/kt/*/*kernel_stats.csv | head -1); cp $KS $O/kernel_stats.csv
KT=$(ls $O/kt/*/*kernel_trace.csv | head -1); python $R/tools/step_trace.py $KT > $O/step_launch_summary.txt 2>&1
# every dispatch of the kernel priced in the bench line's roofline (several shapes share the name)
python - $KT > $O/dominant_kernel_dispatches.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv_big_kernel<bf16, 4, 2, 2, 2, 1>" in r["Kernel_Name"]]
print("start_ns,duration_us,grid,workgroup")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    print("%s,%.1f,%s,%s" % (r["Start_Timestamp"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", ""))))
PY
rm -rf $O/kt
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p1 -- python3 $R/tools/kbench.py conv_t_epi:256:256 --iters 5 > $O/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p2 -- python3 $R/tools/kbench.py conv_t_epi:256:256 --iters 5 > $O/p2.log 2>&1 || exit 1
grep -h "conv_big_kernel\|Correlation_Id" $(ls $O/p1/*/*counter_collection.csv | head -1) > $O/pmc_fetch_size_conv_big.csv
grep -h "conv_big_kernel\|Correlation_Id" $(ls $O/p2/*/*counter_collection.csv | head -1) > $O/pmc_write_size_conv_big.csv
rm -rf $O/p1 $O/p2
timeout -s KILL 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/p3.log 2>&1 || exit 1
python $R/tools/pmc_summary.py $(ls $O/p3/*/*counter_collection.csv | head -1) > $O/pmc_per_kernel_summary.txt
rm -rf $O/p3
echo PROF_DONE
