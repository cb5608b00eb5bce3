#!/bin/bash
# Review item 5, priced by measurement: what would gn_fwd cost if the producing conv's epilogue had left the
# group statistics?  A diagnostic build of norm.hip (-DVF_GN_PRESTATS: mean / rstd read from memory, both block reductions
# gone) against the shipped library, per shape of the small UNet at S = 96, alternating.   bash tools/gn_prestats_ab.sh
cd "$(dirname "$0")/.." || exit 1
mkdir -p build/ab gpurun_out
python -m view_fusion_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -Wno-unused-result -DVF_GN_PRESTATS -c view_fusion_amd/csrc/norm.hip -o build/ab/norm_prestats.o || exit 1
hipcc --offload-arch=gfx950 -shared -fPIC -o build/ab/libvf_gn_prestats.so $(ls build/vf_hip/*.o | grep -v "/norm.o") build/ab/norm_prestats.o || exit 1
for i in 1 2; do
  echo "== shipped"; python tools/gn_table.py 2>/dev/null
  echo "== statistics handed in (diagnostic build)"; VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_gn_prestats.so python tools/gn_table.py 2>/dev/null
done
