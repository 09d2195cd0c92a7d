#!/bin/bash
# One more build of libdm4d_hip.so beside the tree's: tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [file.hip ...]
# recompiles the named translation units with the extra flags (probe switches, -D...), links them with the tree's other objects
# -> build_ab/<name>.so (git-ignored; travels to the GPU box) for tools/ab_many.sh / ab_kernels.sh / prof_two_libs.sh.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift 2
CS=$REPO/dreammesh4d_amd/csrc
make -s -C $CS -j8
mkdir -p $REPO/build_ab/obj_$NAME
OBJS=""
for o in $CS/*.o; do
  b=$(basename $o .o)
  use=$o
  for f in "$@"; do
    if [ "$(basename $f .hip)" == "$b" ]; then
      extra=""
      [ "$b" == "attention" ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I$REPO/include $extra $FLAGS -c $CS/$b.hip -o $REPO/build_ab/obj_$NAME/$b.o
      use=$REPO/build_ab/obj_$NAME/$b.o
    fi
  done
  OBJS="$OBJS $use"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $REPO/build_ab/$NAME.so $OBJS
echo "built build_ab/$NAME.so"
