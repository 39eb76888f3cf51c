# Per-kernel SQ counters of one kbench case: bash tools/pmc_one.sh <case> [dtype] [counters...]  (on the GPU box)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc1; rm -rf $O; mkdir -p $O
CASE=$1; DT=${2:-bf16}; shift; shift
CTRS=${@:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $CTRS --output-format csv -d $O/p -- python3 $R/tools/kbench.py $CASE --dtype $DT --iters 3 > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 $R/tools/pmc_summary.py $(ls $O/p/*/*counter_collection.csv | head -1) 2>/dev/null | grep -v "at::native" | head -3
rm -rf $O/p
