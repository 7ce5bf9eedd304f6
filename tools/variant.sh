#!/bin/bash
# builds cubep3m_amd/libp3m_hip_<tag>.so from the current objects with ONE source recompiled with extra flags (A/B experiments):
#   tools/variant.sh <tag> <source.hip> "<extra hipcc flags>"      run with P3M_HIP_LIB=cubep3m_amd/libp3m_hip_<tag>.so
set -e
cd "$(dirname "$0")/../cubep3m_amd/csrc"
tag=$1; src=$2; extra=$3
make -s -j8
mkdir -p _obj_var
# compiler, architecture, flags and the list of sources come from the Makefile (one place to keep them)
hipcc=$(make -s -pn | sed -n 's/^HIPCC *?*= *//p' | head -1); arch=$(make -s -pn | sed -n 's/^ARCH *?*= *//p' | head -1)
flags=$(make -s -pn | sed -n 's/^CXXFLAGS *= *//p' | head -1 | sed "s/\$(ARCH)/$arch/")
srcs=$(make -s -pn | sed -n 's/^SRCS *= *//p' | head -1)
$hipcc $flags $extra -c $src -o _obj_var/${src%.hip}_$tag.o
objs=""
for f in ${srcs//.hip/}; do
  if [ "$f.hip" == "$src" ]; then objs="$objs _obj_var/${f}_$tag.o"; else objs="$objs _obj/$f.o"; fi
done
$hipcc --offload-arch=$arch -shared -fPIC -o ../libp3m_hip_$tag.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../libp3m_hip_$tag.so
