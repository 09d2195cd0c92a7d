cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${DM4D_PROF_LIBS:-w128 w192}; do
  cp $R/build_ab/$v.so $R/dreammesh4d_amd/libdm4d_hip.so
  rm -rf /tmp/pp_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$v -o p -- python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-iters > /dev/null 2>&1
  echo "== $v"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/pp_$v/**/p_kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:60]:60s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:8.1f}")
PY
done
