#!/bin/bash
# A/B of two builds of libdm4d_hip.so on the SAME GPU box (boxes differ by a few % in clocks):
#   usage (inside gpurun): tools/ab.sh build_ab/old.so build_ab/new.so [rounds]
# prints the top kernels of every run; restores the in-tree library afterwards.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cp $REPO/dreammesh4d_amd/libdm4d_hip.so /tmp/libdm4d_keep.so
for r in $(seq 1 ${3:-2}); do
  for v in $1 $2; do
    cp $REPO/$v $REPO/dreammesh4d_amd/libdm4d_hip.so
    tag=ab_$(basename $v .so)_$r
    bash $REPO/tools/prof_bench.sh $tag
    echo "== $v round $r"; python $REPO/tools/show_prof.py $tag ${4:-5}
  done
done
cp /tmp/libdm4d_keep.so $REPO/dreammesh4d_amd/libdm4d_hip.so
