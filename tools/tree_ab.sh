set -e
mkdir -p gpurun_out
for lib in r03tree base late; do
  echo "=== $lib" 
  AZR_EXP_LIB=alphazero-risk_amd/csrc/dbg/libazr_$lib.so python tools/tree_prof.py --steps 20 2>&1 | grep -v amdgpu.ids
done
