# Diagnostic library with the cycle stamps of the query-chain kernels (tools/attn_timeline.py): tools/abl/libcutie_hip_ATL.so
set -e
cd "$(dirname "$0")/../cutie_amd/csrc"
make -s -j8
mkdir -p ../../tools/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -Wno-unused-value -ffp-contract=off -DATT_TIMELINE -c qchain.hip -o ../../tools/abl/qchain_ATL.o
objs=""; for o in conv_igemm conv_dma conv_pc elementwise stem attention affinity bank api; do [ -f $o.o ] && objs="$objs $o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../tools/abl/qchain_ATL.o -o ../../tools/abl/libcutie_hip_ATL.so
ls -la ../../tools/abl/libcutie_hip_ATL.so
