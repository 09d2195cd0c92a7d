#!/bin/bash
# The launches of ONE steady-state dynamic-stage iteration in order (tools/iters_per_sec.py under rocprofv3 --kernel-trace):
# start offset, duration, gap to the previous launch's end, short kernel name.  Output: $1 (default gpurun_out/iter_sequence.txt)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-gpurun_out/iter_sequence.txt}; case $OUT in /*) ;; *) OUT=$REPO/$OUT ;; esac
mkdir -p $(dirname $OUT)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/its
rocprofv3 --kernel-trace --output-format csv -d /tmp/its -o k -- python $REPO/tools/iters_per_sec.py 2>/dev/null | tail -1
python - "$OUT" <<PY
import csv, glob, re, sys
f = glob.glob('/tmp/its/**/k_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'k_preprocess' in r['Kernel_Name']]
i0, i1 = marks[-3], marks[-2]
t0 = int(rows[i0]['Start_Timestamp']); prev_end = t0
def short(n):
    n = n.replace('void ', '').replace('at::native::', '').replace('(anonymous namespace)::', '')
    m = re.search(r'(CUDAFunctor_\w+|\w+Functor)<([\w:]+)', n)
    if 'elementwise' in n and m: return 'ew ' + m.group(1) + '<' + m.group(2) + '>'
    return n[:110]
with open(sys.argv[1], 'w') as o:
    for r in rows[i0:i1]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        o.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id', '?')} {short(r['Kernel_Name'])}\n")
        prev_end = max(prev_end, e)
print("written", sys.argv[1])
PY
