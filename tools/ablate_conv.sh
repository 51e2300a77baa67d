# Diagnostic: builds three extra libraries whose conv_dma K loop lacks one of its streams (LDS-DMA fill / fragment reads / MFMA)
# into gpurun_out/abl/ (results of those libraries are garbage; only the timing matters).  usage: bash tools/ablate_conv.sh
set -e
cd "$(dirname "$0")/../cutie_amd/csrc"
make -s
mkdir -p ../../tools/abl
# variants: name=macro list
for spec in ${SPECS:-NO_DMA=NO_DMA NO_DSREAD=NO_DSREAD NO_MFMA=NO_MFMA DMA_ONLY=NO_DSREAD,NO_MFMA DMA_ONLY_NOBAR=NO_DSREAD,NO_MFMA,NO_BARRIER EMPTY=NO_DMA,NO_DSREAD,NO_MFMA EMPTY_NOEPI=NO_DMA,NO_DSREAD,NO_MFMA,NO_EPILOGUE EMPTY_NOLOOP=NO_DMA,NO_DSREAD,NO_MFMA,NO_LOOP LAUNCH_ONLY=NO_DMA,NO_DSREAD,NO_MFMA,NO_LOOP,NO_EPILOGUE}; do
  v=${spec%%=*}; defs=""; for m in $(echo ${spec#*=} | tr , ' '); do defs="$defs -DDMA_ABL_$m"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-variable -Wno-unused-value -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
      $defs -c conv_dma.hip -o ../../tools/abl/conv_dma_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC conv_igemm.o conv_pc.o ../../tools/abl/conv_dma_$v.o elementwise.o attention.o qchain.o affinity.o bank.o api.o \
      -o ../../tools/abl/libcutie_hip_$v.so
done
ls -la ../../tools/abl/*.so
