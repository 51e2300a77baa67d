# Round 2, GPU call 1: validate + time the LDS-DMA conv kernel (tiles 60..) and the prepared buffer-load kernel (50..), run the whole
# GPU suite incl. the teacher-forced tests, and time the frame with the swept tile table.  Outputs under gpurun_out/c1/.
OUT=gpurun_out/c1; mkdir -p $OUT
export CUTIE_AMD_EXPERIMENTAL_TILES=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "dma or bufload" > $OUT/1_kernel_tests.log 2>&1; tail -5 $OUT/1_kernel_tests.log
timeout 900 python tools/conv_sweep.py --objects 3 --out $OUT/conv_sweep > $OUT/2_sweep.log 2>&1; tail -3 $OUT/2_sweep.log
unset CUTIE_AMD_EXPERIMENTAL_TILES
timeout 1500 python -m pytest tests -q -m gpu -s > $OUT/3_gpu_suite.log 2>&1; tail -15 $OUT/3_gpu_suite.log | cut -c1-400
CUTIE_AMD_EXPERIMENTAL_TILES=1 CUTIE_AMD_TILE_CACHE=$OUT/conv_sweep_tiles.json timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/4_bench_swept.json 2> $OUT/4_bench_swept.err
timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/4_bench_packaged.json 2> $OUT/4_bench_packaged.err
CUTIE_AMD_EXPERIMENTAL_AFF=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -k "aff" > $OUT/5_aff_tests.log 2>&1; tail -2 $OUT/5_aff_tests.log
CUTIE_AMD_EXPERIMENTAL_AFF=1 timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/5_bench_aff.json 2> $OUT/5_bench_aff.err
python - <<'PY'
import json
for n, f in (('swept tile table', '4_bench_swept'), ('packaged table', '4_bench_packaged'), ('experimental affinity', '5_bench_aff')):
    try:
        d = json.loads(open(f'gpurun_out/c1/{f}.json').read().strip().split('\n')[-1])
        print(n, d['value'], 'frames/s; conv', d.get('roofline', {}).get('achieved'), 'TFLOP/s, frac', d.get('roofline', {}).get('frac'),
              'conv ms/frame', d.get('roofline', {}).get('ms_per_frame'), '; affinity', (d.get('roofline_affinity', {}).get('matmul') or {}).get('stage_us'),
              'mfma_util', (d.get('roofline_affinity', {}).get('matmul') or {}).get('mfma_util'))
    except Exception as e:
        print(n, 'no bench line:', e)
PY
