# rocprofv3 PMC passes for ONE kernel of an arbitrary command (counter collection + kernel trace only).
# usage: PMC_KERNEL=<substring of the kernel name> bash tools/pmc_kernel.sh <command...>   -- prints the LAST matching dispatch;
#        PMC_ALL=1: the MEAN over all matching dispatches instead, and MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCCs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export OMP_NUM_THREADS=16
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_IFETCH SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 280 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmck$i -- "$@" > /tmp/pmck$i.log 2>&1
done
python - <<'PY'
import csv, glob, os
name = os.environ.get('PMC_KERNEL', 'conv')
ALLV = {}
for d in sorted(glob.glob('/tmp/pmck*/')):
    cc = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    kt = glob.glob(d + '**/*kernel_trace.csv', recursive=True)
    if not cc: continue
    rows = [r for r in csv.DictReader(open(cc[0])) if name in r['Kernel_Name']]
    if not rows: print('no dispatch of', name); continue
    last = max(int(r['Dispatch_Id']) for r in rows)
    vals = {r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id']) == last}
    dur = None
    if kt:
        k = sorted([r for r in csv.DictReader(open(kt[0])) if name in r['Kernel_Name']], key=lambda r: int(r['Dispatch_Id']))
        dur = (int(k[-1]['End_Timestamp']) - int(k[-1]['Start_Timestamp'])) / 1e3
        g = (k[-1]['Grid_Size_X'], k[-1]['Grid_Size_Y'], k[-1]['Workgroup_Size_X'], k[-1]['LDS_Block_Size'], k[-1]['VGPR_Count'], k[-1]['Accum_VGPR_Count'])
    if os.environ.get('PMC_ALL'):
        acc = {}
        for r in rows:
            acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        vals = {c: sum(v) / len(v) for c, v in acc.items()}
        n = len(next(iter(acc.values())))
        if kt:
            dur = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in k) / len(k)
        print('mean of', n, 'dispatches, dur_us %.1f' % dur, g, {c: round(v, 1) for c, v in vals.items()})
        ALLV.update(vals)
        continue
    print('dispatch', last, 'dur_us', dur, g, vals)
if ALLV.get('GRBM_GUI_ACTIVE') and ALLV.get('SQ_VALU_MFMA_BUSY_CYCLES'):
    print('MFMA utilisation (busy / SIMD-cycles): %.3f' % (ALLV['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * ALLV['GRBM_GUI_ACTIVE'] / 8)))
    if ALLV.get('SQ_INSTS_VALU_MFMA_MOPS_BF16'):
        print('MOPS_BF16 per dispatch %.0f (x 512 flop)' % ALLV['SQ_INSTS_VALU_MFMA_MOPS_BF16'])
PY
rm -rf /tmp/pmck*
