# Diagnostic libraries (never shipped as the product): tools/abl/libcutie_hip_<NAME>.so = the product library with conv_pc.hip /
# conv_dma.hip rebuilt under extra macros.  usage: bash tools/build_diag.sh [NAME=MACRO,MACRO ...]
#   TL=CONV_TIMELINE                         s_memtime stamps (tools/conv_timeline.py)
#   PC_NO_DMA=PC_ABL_NO_DMA ...              conv_pc ablations (results are garbage; only the timing matters)
set -e
cd "$(dirname "$0")/../cutie_amd/csrc"
make -s -j8
mkdir -p ../../tools/abl
FL="$EXTRA --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -Wno-unused-value -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form"
for spec in ${@:-TL=CONV_TIMELINE}; do
  v=${spec%%=*}; defs=""; for m in $(echo ${spec#*=} | tr , ' '); do defs="$defs -D$m"; done
  /opt/rocm/bin/hipcc $FL $defs -c conv_pc.hip -o ../../tools/abl/conv_pc_$v.o &
  /opt/rocm/bin/hipcc $FL $defs -c conv_dma.hip -o ../../tools/abl/conv_dma_$v.o &
  wait
  objs=""; for o in conv_igemm elementwise stem attention qchain affinity bank api; do objs="$objs $o.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../tools/abl/conv_pc_$v.o ../../tools/abl/conv_dma_$v.o -o ../../tools/abl/libcutie_hip_$v.so
done
ls -la ../../tools/abl/*.so
