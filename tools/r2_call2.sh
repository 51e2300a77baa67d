# Round 2, GPU call 2: why is the LDS-DMA conv kernel not faster?  (1) LDS-DMA fill-rate microbenchmark, (2) PMC of the big 3x3 layer.
OUT=gpurun_out/c2; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/dmabench.hip -o /tmp/dmabench 2> $OUT/dmabench_build.log && timeout 120 /tmp/dmabench > $OUT/1_dmabench.txt 2>&1; tail -5 $OUT/1_dmabench.txt
for t in 60 66 61 63; do
  PMC_KERNEL=conv_dma bash tools/pmc_kernel.sh python tools/one_conv.py 1 360 216 128 128 3 $t 10 > $OUT/2_pmc_77760_t$t.txt 2>&1
done
for t in 67 60; do
  PMC_KERNEL=conv_dma bash tools/pmc_kernel.sh python tools/one_conv.py 3 30 54 256 256 3 $t 10 > $OUT/2_pmc_4860_t$t.txt 2>&1
done
PMC_KERNEL=conv_igemm bash tools/pmc_kernel.sh python tools/one_conv.py 1 360 216 128 128 3 14 10 > $OUT/2_pmc_77760_t14.txt 2>&1
tail -2 $OUT/2_pmc_77760_t60.txt | cut -c1-300
