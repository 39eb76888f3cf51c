# HBM-side traffic of single kernels at the training step's shapes, from the memory-side counters: one rocprofv3 run per
# (kernel case, counter), FETCH_SIZE and WRITE_SIZE in separate passes.  Run on the GPU box as `bash tools/pmc_traffic.sh`;
# the summary (gpurun_out/prof/traffic.json) is merged into bench_traffic.json with `python tools/pmc_traffic.py --merge`.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof/traffic; rm -rf $O; mkdir -p $O   # keeps the other files of gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "bf16 conv_t_epi:256:256" "bf16 conv:256:256" "bf16 conv_t_epi:128:128" "bf16 conv:128:128" "bf16 conv_t_epi:64:64" "bf16 conv:64:64" \
            "bf16 conv_wgrad:256:256" "bf16 conv_wgrad:64:64" \
            "bf16 gcn_fused:64:64" "bf16 gcn_fused_add:64:64" "bf16 gcn_fused:256:256" "bf16 gcn_fused_add:256:256" \
            "bf16 gcn_wgrad:64:64" "bf16 gcn_agrad:64:64" "bf16 gcn_wgrad:256:256" "bf16 gcn_agrad:256:256" \
            "bf16 bnbwd:64:64" "bf16 bofwd:64:64" "bf16 bob0:64:64" "bf16 bob1:64:64" "bf16 bnbwd:256:256" "bf16 bob1:256:256" \
            "bf16 gcn_fused:128:128" "bf16 gcn_fused_add:128:128" "bf16 gcn_wgrad:128:128" "bf16 gcn_agrad:128:128" "bf16 conv_wgrad:128:128" \
            "bf16 bofwd:256:256" "bf16 bob0:256:256" "bf16 bob1g:256:256" \
            "fp32 conv_wgrad:256:256" "fp32 conv_t_epi:256:256" "fp32 conv:256:256" "fp32 gcn_fused:64:64" "fp32 bnbwd:64:64"; do
  set -- $spec
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $O/c${i}_$ctr -- python3 $R/tools/kbench.py $2 --dtype $1 --iters 3 --print-key > $O/c${i}_$ctr.log 2>&1 || { echo "FAILED $spec $ctr"; tail -5 $O/c${i}_$ctr.log; exit 1; }
    echo "done $spec $ctr"
  done
  i=$((i+1))
done
python3 $R/tools/pmc_traffic.py $O > $R/gpurun_out/prof/traffic.json && rm -rf $O/*/ && echo TRAFFIC_DONE
