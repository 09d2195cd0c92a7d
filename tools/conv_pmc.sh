#!/bin/bash
# SQ counters of the conv kernel for one shape and configuration: tools/conv_pmc.sh <cfg> N H Cin Cout
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=$1; shift
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/cpmc
  DM4D_CONV_CFG=$CFG rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/cpmc -o p -- python $R/tools/conv_one.py "$@" > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/cpmc/**/p_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_conv3x3' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()): print(f"   cfg $CFG  {k:28s} {sum(v)/len(v):16.0f}")
PY
done
