# round 3, GPU call 2: conv_pc with deeper DMA lead + register double-buffered fragments + one-batch kernel arguments
cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r3c2
mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "test_conv_pc_tiles and (0-100 or 0-120)" > $O/0_canary.log 2>&1
echo "canary rc=$?" >> $O/0_canary.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu --maxfail=40 -k "conv_pc or gap_accum" > $O/1_tests.log 2>&1
CUTIE_AMD_LIB=tools/abl/libcutie_hip_TL.so timeout 400 python tools/conv_timeline.py --tiles 66 100 101 107 120 123 > $O/3_timeline.log 2>&1
timeout 1500 python tools/conv_sweep.py --objects 3 --families dma,pc,halo --out $O/conv_sweep > $O/4_sweep.log 2>&1
for v in 0 1; do HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --steps 60 --warmup 10 > $O/5_bench_devkernarg$v.json 2> $O/5_bench_devkernarg$v.err; done
tail -n 3 $O/0_canary.log $O/1_tests.log
tail -n 50 $O/4_sweep.log
