#!/bin/bash
# Runs ON THE GPU BOX: Zero123 SDS step numbers + rocprofv3 kernel stats + MFMA counters -> gpurun_out/zero123_<tag>/
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/zero123_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/zero123_profile.py > $OUT/summary.json 2> $OUT/summary.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/z123s -o z -- python $REPO/tools/zero123_profile.py --steps-only 10 > $OUT/stats.log 2>&1
cp /tmp/z123s/*/z_kernel_stats.csv /tmp/z123s/z_kernel_stats.csv $OUT/ 2>/dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d /tmp/z123p -o z -- python $REPO/tools/zero123_profile.py --steps-only 3 > $OUT/pmc.log 2>&1
python - <<PY
import csv, glob, collections, json
f = glob.glob('/tmp/z123p/**/z_counter_collection.csv', recursive=True)
acc = collections.Counter()
if f:
    for r in csv.DictReader(open(f[0])):
        acc[r['Counter_Name']] += float(r['Counter_Value'])
json.dump(dict(acc), open('$OUT/pmc_totals.json', 'w'))
print(dict(acc))
PY
cat $OUT/summary.json; ls $OUT
