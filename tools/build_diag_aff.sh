# Diagnostic library with the affinity timeline (never shipped): tools/abl/libcutie_hip_ATL.so = the product library with affinity.hip
# rebuilt under -DAFF_TIMELINE (tools/aff_timeline.py).
set -e
cd "$(dirname "$0")/../cutie_amd/csrc"
make -s -j8
mkdir -p ../../tools/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -Wno-unused-value -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DAFF_TIMELINE -c affinity.hip -o ../../tools/abl/affinity_ATL.o
objs=""; for o in conv_igemm conv_dma conv_pc elementwise stem attention qchain bank api; do objs="$objs $o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../tools/abl/affinity_ATL.o -o ../../tools/abl/libcutie_hip_ATL.so
ls -la ../../tools/abl/libcutie_hip_ATL.so
