#!/bin/bash
# builds cubep3m_amd/libp3m_hip_<tag>.so from the current objects with ONE source recompiled with extra flags (A/B experiments):
#   tools/variant.sh <tag> <source.hip> "<extra hipcc flags>"      run with P3M_HIP_LIB=cubep3m_amd/libp3m_hip_<tag>.so
# The compile line of the object is make's own (per-file rules included: fft.hip and kick_fused.hip without -ffp-contract=off, pp.hip with
# -fno-slp-vectorize), with the output path replaced and the extra flags appended.
set -e
cd "$(dirname "$0")/../cubep3m_amd/csrc"
tag=$1; src=$2; extra=$3
make -s -j8
mkdir -p _obj_var
obj=_obj/${src%.hip}.o
line=$(make -n -W $src $obj | grep -- "-c $src" | head -1)
[ -n "$line" ] || { echo "no compile line for $src"; exit 1; }
line=${line/-o $obj/-o _obj_var/${src%.hip}_$tag.o}
eval "$line $extra"
srcs=$(make -s -pn | sed -n 's/^SRCS *= *//p' | head -1)
hipcc=$(make -s -pn | sed -n 's/^HIPCC *?*= *//p' | head -1); arch=$(make -s -pn | sed -n 's/^ARCH *?*= *//p' | head -1)
objs=""
for f in ${srcs//.hip/}; do
  if [ "$f.hip" == "$src" ]; then objs="$objs _obj_var/${f}_$tag.o"; else objs="$objs _obj/$f.o"; fi
done
$hipcc --offload-arch=$arch -shared -fPIC -o ../libp3m_hip_$tag.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../libp3m_hip_$tag.so
