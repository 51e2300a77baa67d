"""Condense the rocprofv3 output of tools/profile_round.sh into profiles/<round>_*.{csv,json}."""
import collections, csv, glob, json, os, re, sys

src, rnd = sys.argv[1], sys.argv[2]
os.makedirs('profiles', exist_ok=True)


def short(name):
    name = re.sub(r'^void ', '', name)
    m = re.match(r'(conv_igemm_kernel|conv_dma_kernel|conv_pc_kernel)<', name)
    if m:
        return m.group(1) + '<*>'
    m = re.match(r'(\w+)<[^>]*>', name)
    return (m.group(1) + '<*>') if m and name.startswith(('aff_score', 'linear_small')) else name.split('(')[0][:90]


st = glob.glob(os.path.join(src, 'stats', '**', '*kernel_stats.csv'), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(f'profiles/{rnd}_bench_kernel_stats.csv', 'w') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
        for r in rows:
            w.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        a = agg[short(r['Name'])]
        a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
    tot = sum(v[1] for v in agg.values())
    fam = [{'kernel': k, 'calls': v[0], 'total_ms': round(v[1] / 1e6, 3), 'avg_us': round(v[1] / v[0] / 1e3, 2), 'pct': round(100 * v[1] / tot, 2)}
           for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])]
else:
    fam = []
# per-stream occupancy of the steady state (last 40 % of the dispatches): which HIP stream is the critical path of a frame
streams = {}
kt = glob.glob(os.path.join(src, 'stats', '**', '*kernel_trace.csv'), recursive=True)
if kt:
    tr = list(csv.DictReader(open(kt[0])))
    tr.sort(key=lambda r: int(r['Start_Timestamp']))
    tr = tr[int(len(tr) * 0.6):]
    key = 'Stream_Id' if tr and 'Stream_Id' in tr[0] else 'Queue_Id'
    span = (int(tr[-1]['End_Timestamp']) - int(tr[0]['Start_Timestamp'])) / 1e6 if tr else 0.0
    per = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for r in tr:
        a = per[r[key]]
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        a[0] += 1; a[1] += d; a[2][short(r['Kernel_Name'])] += d
    streams = {'keyed_by': key, 'span_ms': round(span, 3),
               'streams': {k: {'dispatches': v[0], 'busy_ms': round(v[1], 3), 'busy_over_span': round(v[1] / span, 3) if span else None,
                               'top': [[n, round(t, 3)] for n, t in v[2].most_common(6)]} for k, v in per.items()}}
pm = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in glob.glob(os.path.join(src, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        pm[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
kern = {}
for k, c in pm.items():
    d = dict(c)
    n = {cn: len(disp[(k, cn)]) for cn in c}
    out = {cn: v for cn, v in d.items()}
    out['dispatches'] = max(n.values())
    if 'FETCH_SIZE' in d:
        # FETCH_SIZE / WRITE_SIZE are in KiB-like 1 KB units; gfx950 reports 1/2 for wide coalesced reads (MI355X_MICROARCH.md)
        out['fetch_MB_per_dispatch_x2_gfx950_corrected'] = round(2 * d['FETCH_SIZE'] / n['FETCH_SIZE'] / 1024, 3)
    if 'WRITE_SIZE' in d:
        out['write_MB_per_dispatch'] = round(d['WRITE_SIZE'] / n['WRITE_SIZE'] / 1024, 3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in d and d.get('GRBM_GUI_ACTIVE'):
        out['mfma_busy_over_gui_active'] = round(d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['GRBM_GUI_ACTIVE'], 3)
    if d.get('SQ_LDS_IDX_ACTIVE'):
        out['lds_bank_conflict_frac'] = round(d.get('SQ_LDS_BANK_CONFLICT', 0.0) / d['SQ_LDS_IDX_ACTIVE'], 4)
    kern[k] = out
# ---- the score pass of the affinity read-out on the HINTED leg, per kernel instantiation: counters next to arithmetic
# SQ_VALU_MFMA_BUSY_CYCLES counts 16 cycles per v_mfma_f32_16x16x32_bf16 (= MOPS / 2: r05 summary; MI355X_MICROARCH.md: 32 per 32x32x16), a SIMD
# issues one per 16 cycles: utilisation by counter = busy cycles / (1024 SIMDs x cycles of the dispatch); GRBM_GUI_ACTIVE is summed over the 8 XCCs.
hinted = {}
hc = glob.glob(os.path.join(src, 'hinted', '**', '*counter_collection.csv'), recursive=True)
if hc:
    per = collections.defaultdict(lambda: collections.defaultdict(dict))       # kernel -> dispatch -> counter -> value
    dur = {}
    for r in csv.DictReader(open(hc[0])):
        name = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0]
        if not name.startswith('aff_score'):
            continue
        per[name][r['Dispatch_Id']][r['Counter_Name']] = per[name][r['Dispatch_Id']].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if r.get('Start_Timestamp') and r.get('End_Timestamp'):
            dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if not dur:
        for f in glob.glob(os.path.join(src, 'hinted', '**', '*kernel_trace.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r.get('Dispatch_Id', r.get('Correlation_Id'))] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for name, ds in per.items():
        n = len(ds)
        tot = collections.defaultdict(float)
        ns = 0
        for did, c in ds.items():
            for k, v in c.items():
                tot[k] += v
            ns += dur.get(did, 0)
        cyc = tot.get('GRBM_GUI_ACTIVE', 0.0) / 8
        o = {'dispatches': n, 'avg_us_profiled': round(ns / n / 1e3, 2) if ns else None,
             'mfma_busy_cycles_per_dispatch': round(tot.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / n), 'mfma_mops_bf16_per_dispatch': round(tot.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) / n),
             'gui_active_cycles_per_xcc_per_dispatch': round(cyc / n)}
        if cyc:
            o['mfma_util_by_counter'] = round(tot.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024 * cyc), 4)      # busy SIMD-cycles / SIMD-cycles of the dispatches
        if ns:
            o['clock_ghz_profiled'] = round(cyc / ns, 3) if cyc else None
            o['issued_tflops_by_counter'] = round(tot.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) * 512 / ns / 1e3, 1)       # MOPS x 512 flop / time
            o['mfma_util_flops_over_time'] = round(tot.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) * 512 / ns / 1e3 / 2500.0, 4)
            if cyc:     # the same utilisation at the 2.4 GHz the 2.5 PFLOP/s peak is quoted at: counter figure x (profiled clock / 2.4)
                o['mfma_util_by_counter_at_peak_clock'] = round(o['mfma_util_by_counter'] * (cyc / ns) / 2.4, 4)
        hinted[name] = o
json.dump({'tree': os.environ.get('CUTIE_TREE', 'unknown'),        # git revision the snapshot was taken from (the GPU box has no .git: passed in by the caller)
           'command': 'rocprofv3 --kernel-trace --stats / --pmc <set> --kernel-trace -- python bench.py --steps 60 --warmup 10 --preroll 60 '
                      '--cpu-frames 0 --no-roofline --clips-in-flight 0 (tools/profile_round.sh; separate passes for the SQ set, FETCH_SIZE, WRITE_SIZE)',
           'notes': 'sums over every dispatch of the run (incl. the conv autotune trials at the first frames); GRBM_GUI_ACTIVE is summed over the 8 XCCs',
           'families_by_time': fam, 'streams_steady_state': streams, 'pmc': kern,
           'affinity_score_pass_hinted': {'command': 'the SQ set on the same bench command WITH look-ahead hints (stacked read-outs)', 'kernels': hinted}}, open(f'profiles/{rnd}_summary.json', 'w'), indent=1)
print(json.dumps(fam[:12], indent=0))
print(json.dumps(streams, indent=0)[:3000])
