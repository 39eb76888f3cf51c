# Runs one whole-model gradient case with two builds of the library / two settings: prints the number of ReLU decisions
# that differ from float64 and the direct deviation.
cd $GRAFT_REPO_ROOT
for lib in "" tools/ab/libstgcn_oldshift.so; do
  STGCN_LIB=$lib timeout -k 10 120 python -u -m pytest tests/test_gpu_model.py -q -m gpu -s -k "train and f32 and test_model" --timeout 100 --timeout-method thread 2>&1 | grep -E "^ReLU decisions|^worst gradient|passed|failed" | cut -c1-200 | sed "s|^|lib=$lib: |"
done
