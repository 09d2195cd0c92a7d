cd /tmp && export TMPDIR=/tmp
for v in bl384 bl8; do
cp /root/repo/build_ab/$v.so /root/repo/dreammesh4d_amd/libdm4d_hip.so
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_lo_$v -o bench -- python /root/repo/tools/long_only.py > /root/repo/gpurun_out/prof_lo_$v.log 2>&1
echo "== $v"; python /root/repo/tools/show_prof.py lo_$v 40 | grep "render_bwd"
done
