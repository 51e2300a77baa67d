# Round 2, GPU call 3: is the operand stream of the conv kernels L2-channel hot-spotted (all blocks streaming the same weight tiles in lockstep)?
OUT=gpurun_out/c3; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/dmabench.hip -o /tmp/dmabench 2> $OUT/dmabench_build.log && timeout 120 /tmp/dmabench > $OUT/1_dmabench.txt 2>&1; tail -3 $OUT/1_dmabench.txt
for kst in 0 5 7; do
  for cfg in "1 360 216 128 128 3 60" "1 360 216 128 128 3 66" "1 360 216 128 128 3 63" "3 30 54 256 256 3 67" "3 30 54 256 256 3 63" "1 30 54 1024 256 1 67" "1 30 54 256 1024 1 65" "3 30 54 512 768 3 63"; do
    echo "kstag $kst cfg $cfg: $(CUTIE_DMA_KSTAG=$kst python tools/one_conv.py $cfg 5 2>&1 | tail -1)" >> $OUT/2_kstag.txt
  done
done
cat $OUT/2_kstag.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kst in 0 5; do
CUTIE_DMA_KSTAG=$kst timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d /tmp/pmcA$kst -- python tools/one_conv.py 1 360 216 128 128 3 60 5 > /tmp/pmcA.log 2>&1
CUTIE_DMA_KSTAG=$kst timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d /tmp/pmcB$kst -- python tools/one_conv.py 1 360 216 128 128 3 60 5 > /tmp/pmcB.log 2>&1
done
python - <<'PY' > gpurun_out/c3/3_tcc.txt 2>&1
import csv, glob
for d in sorted(glob.glob('/tmp/pmc[AB]*/')):
    cc = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    if not cc: print(d, 'no counters'); continue
    rows = [r for r in csv.DictReader(open(cc[0])) if 'conv_dma' in r['Kernel_Name']]
    if not rows: print(d, 'no conv_dma dispatch'); continue
    last = max(int(r['Dispatch_Id']) for r in rows)
    print(d, {r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id']) == last})
PY
cat gpurun_out/c3/3_tcc.txt
