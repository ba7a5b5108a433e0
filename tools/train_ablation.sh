export TMPDIR=/tmp
for v in ${VARIANTS:-product nostash noepi both}; do
  L=""; [ $v != product ] && L=alphazero-risk_amd/csrc/dbg/libazr_$v.so
  AZR_EXP_LIB=$L bash tools/profile_train.sh abl_$v --batches 4 --epochs 2 > /dev/null 2>&1
  echo "== $v"; python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_train_abl_$v/kernel_stats.csv")))
for r in rows:
    if "t_conv" in r["Name"] or "t_wgrad" in r["Name"]: print(r["Name"][28:62], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
