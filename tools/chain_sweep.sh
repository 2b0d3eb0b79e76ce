#!/bin/bash
# measurement harness: the chain kernel's variants (tools/chain_probe.py) under rocprofv3, for a list of "VEC BPC" settings
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/chain_sweep
for cfg in "${@:-4 0}"; do
  set -- $cfg
  GRB_MI355X_CHAIN_VEC=$1 GRB_MI355X_CHAIN_BPC=$2 timeout 100 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/chain_sweep/p_$1_$2 -o x -- python tools/chain_probe.py > /dev/null 2>&1
  f=$(find gpurun_out/chain_sweep/p_$1_$2 -name "*kernel_trace.csv" | head -1)
  python - $f "$cfg" <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_vec_chain" in r["Kernel_Name"]]
out=[]
for i,r in enumerate(rows):
    if i%20==10: out.append("%s %.1f" % (r["Kernel_Name"].split("(")[0][-28:], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
print("VEC,BPC =", sys.argv[2], "|", " | ".join(out))
PY
  rm -rf gpurun_out/chain_sweep/p_$1_$2
done
