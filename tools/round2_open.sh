# Opening sequence of round 2 (one gpurun call, ~6-8 GPU-minutes): validate and time the prepared buffer-load conv kernel.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/round2_open.sh'
# 1. parity of tiles 50-56 against the descriptor interpreter (63 cases);  2. the whole conv family still agrees on the real
# layers with the experimental tiles among the candidates;  3. re-tune every conv geometry with them enabled and report which
# layers switch;  4. bench with the new table vs the packaged one.  Outputs under gpurun_out/round2_open/.
export CUTIE_AMD_EXPERIMENTAL_TILES=1
OUT=gpurun_out/round2_open; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k bufload > $OUT/1_bufload_tests.log 2>&1; tail -3 $OUT/1_bufload_tests.log
if ! grep -q " passed" $OUT/1_bufload_tests.log || grep -q "failed" $OUT/1_bufload_tests.log; then echo "bufload tests not green: stop here"; exit 1; fi
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "every_conv_candidate" > $OUT/2_candidates.log 2>&1; tail -2 $OUT/2_candidates.log
CUTIE_AMD_TUNE=5x16 CUTIE_AMD_TILE_CACHE=$OUT/tiles_experimental.json timeout 600 python tools/tune_tiles.py > $OUT/3_tune.log 2>&1; tail -2 $OUT/3_tune.log
python - <<'PY'
import json
load = lambda f: {tuple(k): tuple(v) for k, v in json.load(open(f))['tiles']}
new, old = load('gpurun_out/round2_open/tiles_experimental.json'), load('cutie_amd/tiles_gfx950.json')
sw = {k: (old.get(k), v) for k, v in new.items() if v[0] >= 50}
print(len(sw), 'of', len(new), 'conv geometries now pick an experimental tile (key = M, Cout, Cin, k, stride, flags&3, H, W)')
for k, v in sorted(sw.items()):
    print('  ', k, 'packaged', v[0], '-> now', v[1])
PY
CUTIE_AMD_TILE_CACHE=$OUT/tiles_experimental.json timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/4_bench_experimental.json 2> $OUT/4_bench_experimental.err
unset CUTIE_AMD_EXPERIMENTAL_TILES
timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/4_bench_packaged.json 2> $OUT/4_bench_packaged.err
# 5. the compile-time-pass affinity score kernels (affinity.hip, AFF_MODE 0 / 1): parity, then the matmul's MFMA utilisation
CUTIE_AMD_EXPERIMENTAL_AFF=1 timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -q -x -k "aff or trajectory" > $OUT/5_aff_tests.log 2>&1; tail -2 $OUT/5_aff_tests.log
CUTIE_AMD_EXPERIMENTAL_AFF=1 timeout 300 python bench.py --steps 200 --warmup 20 --cpu-frames 0 --clips-in-flight 0 > $OUT/5_bench_aff.json 2> $OUT/5_bench_aff.err
python - <<'PY'
import json
for n, f in (('experimental conv tiles', '4_bench_experimental'), ('packaged', '4_bench_packaged'), ('experimental affinity', '5_bench_aff')):
    try:
        d = json.loads(open(f'gpurun_out/round2_open/{f}.json').read().strip().split('\n')[-1])
        print(n, d['value'], 'frames/s; conv', d.get('roofline', {}).get('achieved'), 'TFLOP/s, frac', d.get('roofline', {}).get('frac'),
              '; affinity matmul mfma_util', (d.get('roofline_affinity', {}).get('matmul') or {}).get('mfma_util'))
    except Exception as e:
        print(n, 'no bench line:', e)
PY
