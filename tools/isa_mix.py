"""Instruction mix of the hottest loop of every kernel, from the gfx950 assembly hipcc emits (no GPU needed).

For each kernel the loop body (label ... backward branch to that label) that holds the most MFMAs -- or, for kernels without
MFMA, the most instructions -- is classified into MFMA / VALU / SALU / LDS (ds_*) / VMEM (global/buffer/flat/scratch) / waitcnt /
barrier / branch.  `non-MFMA/MFMA` = the other instructions per MFMA in that loop.  Reading aid (constants of MI355X_MICROARCH.md): a wave issues
its instructions in order; a wave64 VALU instruction takes 2 cycles on a SIMD-32, a v_mfma_f32_16x16x32_bf16 keeps the matrix
pipe of its SIMD busy for ~16-17 cycles.  A loop with R non-MFMA instructions per MFMA therefore needs >= ~R (+VALU) issue cycles
of its wave per 16-17 cycles of matrix work: above R ~ 8 a single wave cannot keep the matrix pipe fed and the kernel relies on
several resident waves interleaving; when VALU x 2 alone exceeds MFMA x 16 the vector pipe is the bound.

    python tools/isa_mix.py [file.hip ...] > profiles/rNN_isa_mix.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
DEFAULT = ['conv_igemm', 'conv_dma', 'conv_pc', 'affinity', 'attention', 'qchain', 'elementwise', 'bank']


def classify(op):
    if op.startswith('v_mfma') or op.startswith('v_smfmac'):
        return 'mfma'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('v_'):
        return 'valu'
    return 'other'


def kernels_of(asm):
    """name -> list of (label or None, opcode) in program order."""
    out, cur, name = {}, None, None
    for line in asm.split('\n'):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is None:
            continue
        if re.match(r'^\s*\.(end_amdhsa_kernel|size)\b', line) or line.startswith('.Lfunc_end'):
            cur = None if line.startswith('.Lfunc_end') else cur
            continue
        m = re.match(r'^(\.LBB\w+):', line)
        if m:
            cur.append((m.group(1), None))
            continue
        m = re.match(r'^\s+([a-z_0-9]+)\b(.*)$', line)
        if m and not m.group(1).startswith('.'):
            cur.append((None, (m.group(1), m.group(2))))
    return out


def hottest_loop(items):
    labels = {lab: i for i, (lab, op) in enumerate(items) if lab}
    best = None
    for i, (lab, op) in enumerate(items):
        if op and op[0].startswith(('s_cbranch', 's_branch')):
            m = re.search(r'(\.LBB\w+)', op[1])
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                body = [o[0] for (l, o) in items[labels[m.group(1)]:i + 1] if o]
                n_mfma = sum(1 for o in body if classify(o) == 'mfma')
                key = (n_mfma, len(body))
                # innermost preference: among loops with the same MFMA count take the shortest
                if best is None or key[0] > best[0][0] or (key[0] == best[0][0] and key[0] > 0 and key[1] < best[0][1]) \
                        or (key[0] == best[0][0] == 0 and key[1] > best[0][1]):
                    best = (key, body)
    return best[1] if best else []


def main():
    files = [a for a in sys.argv[1:]] or DEFAULT
    names, rows = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            src = f if f.endswith('.hip') else os.path.join(ROOT, 'cutie_amd', 'csrc', f + '.hip')
            out = os.path.join(tmp, os.path.basename(src) + '.s')
            subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                            '-I' + os.path.join(ROOT, 'include'), '-S', '--cuda-device-only', '-o', out, src],
                           check=True, capture_output=True)
            for name, items in kernels_of(open(out).read()).items():
                body = hottest_loop(items)
                mix = {}
                for o in body:
                    mix[classify(o)] = mix.get(classify(o), 0) + 1
                total = sum(1 for (l, o) in items if o)
                names.append(name)
                rows.append((os.path.basename(src)[:-4], mix, len(body), total))
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.strip().split('\n')
    cols = ['mfma', 'valu', 'salu', 'lds', 'vmem', 'wait', 'barrier', 'branch', 'nop']
    print(f'{"kernel (hottest loop)":<84s} ' + ' '.join(f'{c:>5s}' for c in cols) + '  loop  total  non-MFMA/MFMA')
    for n, (f, mix, nb, total) in zip(dem, rows):
        n = re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', n))
        ratio = (nb - mix.get('mfma', 0)) / mix['mfma'] if mix.get('mfma') else float('nan')
        print(f'{(f + ": " + n)[:84]:<84s} ' + ' '.join(f'{mix.get(c, 0):5d}' for c in cols) + f' {nb:5d} {total:6d}  {ratio:8.2f}')


if __name__ == '__main__':
    main()
