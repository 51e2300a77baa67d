"""Static resource usage of every gfx950 kernel in libcutie_hip.so (no GPU needed): VGPRs / AGPRs / SGPRs, scratch (spills),
LDS bytes and the occupancy the compiler derives from them -- hipcc's -Rpass-analysis=kernel-resource-usage remarks folded into
one table.    python tools/kernel_resources.py > profiles/r01_kernel_resources.txt"""
import os
import re
import subprocess
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
SRCS = ['conv_igemm', 'conv_dma', 'conv_pc', 'elementwise', 'attention', 'qchain', 'affinity', 'bank']
KEYS = ['VGPRs', 'AGPRs', 'TotalSGPRs', 'ScratchSize [bytes/lane]', 'VGPRs Spill', 'SGPRs Spill', 'LDS Size [bytes/block]',
        'Occupancy [waves/SIMD]']


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
    return out.strip().split('\n')


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in SRCS:
            r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
                                '-I' + os.path.join(ROOT, 'include'), '-c', os.path.join(ROOT, 'cutie_amd', 'csrc', src + '.hip'),
                                '-o', os.path.join(tmp, src + '.o'), '-Rpass-analysis=kernel-resource-usage'],
                               capture_output=True, text=True)
            cur = None
            for line in r.stderr.split('\n'):
                m = re.search(r'remark:\s+Function Name: (\S+)', line)
                if m:
                    cur = {'file': src, 'name': m.group(1)}
                    rows.append(cur)
                    continue
                m = re.search(r'remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass', line)
                if m and cur is not None:
                    cur[m.group(1).strip()] = m.group(2)
    names = demangle([r['name'] for r in rows])
    print(f'{"kernel":<96s} {"VGPR":>4s} {"AGPR":>4s} {"SGPR":>4s} {"scratch":>7s} {"vspill":>6s} {"LDS B":>7s} {"occ":>3s}')
    for r, n in zip(rows, names):
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'\(.*\)$', '', n)
        lds = r.get('LDS Size [bytes/block]', '?')
        print(f'{(r["file"] + ": " + n)[:96]:<96s} {r.get("VGPRs", "?"):>4s} {r.get("AGPRs", "?"):>4s} {r.get("TotalSGPRs", "?"):>4s} '
              f'{r.get("ScratchSize [bytes/lane]", "?"):>7s} {r.get("VGPRs Spill", "?"):>6s} {lds:>7s} {r.get("Occupancy [waves/SIMD]", "?"):>3s}')
    spills = [n for r, n in zip(rows, names) if r.get('ScratchSize [bytes/lane]', '0') != '0']
    print(f'\n{len(rows)} kernels; {len(spills)} with scratch: {spills}')
    print('(LDS of the conv kernels is dynamic (cutie_amd/csrc/conv_igemm.hip conv_lds_bytes): the static figure is 0 there.)')


if __name__ == '__main__':
    main()
