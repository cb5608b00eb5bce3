#!/bin/bash
# Diagnostic builds of winograd44f.hip with one kind of chunk-loop side work left out (results are WRONG by
# construction; only the in-kernel cycle stamps are read): run on the GPU box after `bash tools/wino44f_ablate.sh build`.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ab
OBJS=$(ls build/vf_hip/*.o | grep -v winograd44f)
if [ "$1" = build ]; then
  for v in BASE NOXFORM NOX NOU NOBF ALL; do
    F="-DVF_STAMPS44F"
    [ $v = ALL ] && F="$F -DVF_AB_NOXFORM -DVF_AB_NOX -DVF_AB_NOU -DVF_AB_NOBF"
    [ $v != BASE ] && [ $v != ALL ] && F="$F -DVF_AB_$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -Wno-unused-result $F -c view_fusion_amd/csrc/winograd44f.hip -o build/ab/w44f_$v.o &
  done
  wait
  for v in BASE NOXFORM NOX NOU NOBF ALL; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libvf_w44f_$v.so $OBJS build/ab/w44f_$v.o
  done
  exit 0
fi
for v in BASE NOXFORM NOX NOU NOBF ALL; do
  echo "== $v"
  VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_w44f_$v.so python tools/wino44f_stamps.py 2 2>&1 | grep "^Cin"
done
