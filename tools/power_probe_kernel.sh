# Shader clock while ONE kernel runs back to back for a few seconds (tools/kbench.py case given as $1, iterations $2):
# an MFMA-dense stream against a memory-bound one.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/power; mkdir -p $O
cd $R
CASE=${1:-conv_t_epi:256:256}; IT=${2:-15000}
(timeout -k 5 120 python tools/kbench.py $CASE --iters $IT > $O/kb_$$.txt 2>&1; echo KB_DONE >> $O/kb_$$.txt) &
BP=$!
: > $O/clk_$$.txt
for i in $(seq 1 120); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  timeout 5 rocm-smi --showclocks --showuse 2>&1 | grep -E "sclk|GPU use" | awk '{print $NF}' | tr '\n' ' ' >> $O/clk_$$.txt
  echo >> $O/clk_$$.txt
  sleep 0.15
done
wait $BP
echo "case $CASE"; grep -v amdgpu.ids $O/kb_$$.txt | head -3
awk '$2+0 >= 95 {gsub(/[()Mhz]/,"",$1); s+=$1; n++} END {if (n) printf "sclk while busy: mean %.0f MHz over %d samples\n", s/n, n}' $O/clk_$$.txt
