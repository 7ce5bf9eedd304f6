#!/bin/bash
# builds cubep3m_amd/libp3m_hip_<tag>.so from the current objects with ONE source recompiled with extra flags (A/B experiments):
#   tools/variant.sh <tag> <source.hip> "<extra hipcc flags>"      run with P3M_HIP_LIB=cubep3m_amd/libp3m_hip_<tag>.so
set -e
cd "$(dirname "$0")/../cubep3m_amd/csrc"
tag=$1; src=$2; extra=$3
make -s -j8
mkdir -p _obj_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -ffp-contract=off $extra -c $src -o _obj_var/${src%.hip}_$tag.o
objs=""
for f in p3m_api fft scan particles fine_mesh pp coarse_mesh group timestep io_formats; do
  if [ "$f.hip" == "$src" ]; then objs="$objs _obj_var/${f}_$tag.o"; else objs="$objs _obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libp3m_hip_$tag.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built ../libp3m_hip_$tag.so
