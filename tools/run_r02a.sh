# round-2 first GPU pass: SQ counters of the r01 kernels (baseline), cfg5 stress, Zero123 probe + kernel stats
mkdir -p gpurun_out/r02a
bash tools/pmc_sq.sh > gpurun_out/r02a/pmc_sq.log 2>&1
python tools/pmc_sq.py render_fwd render_bwd tile_sort gather_bwd > gpurun_out/r02a/pmc_sq_summary.txt 2>&1
rm -rf gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/pmc_sq_c
timeout 600 python tools/stress_cfg5.py > gpurun_out/r02a/stress_cfg5.log 2>&1
timeout 600 python tools/zero123_probe.py > gpurun_out/r02a/zero123_probe.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/z123 -o z -- python /root/repo/tools/zero123_prof.py > /root/repo/gpurun_out/r02a/z123.log 2>&1
mkdir -p /root/repo/gpurun_out/r02a/z123; cp /tmp/z123/*/*stats*.csv /tmp/z123/*stats*.csv /root/repo/gpurun_out/r02a/z123/ 2>/dev/null
cd /root/repo; du -sh gpurun_out; tail -5 gpurun_out/r02a/*.log; head -50 gpurun_out/r02a/pmc_sq_summary.txt
