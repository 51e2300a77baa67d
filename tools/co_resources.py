"""VGPR / SGPR / scratch / LDS of every kernel in an object file built for gfx950 (no GPU, no recompile): reads the code object's
metadata notes.    python tools/co_resources.py cutie_amd/csrc/conv_pc.o [name filter]"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = '/opt/rocm/lib/llvm/bin/'


def kernels(obj):
    with tempfile.TemporaryDirectory() as tmp:
        fb, co = os.path.join(tmp, 'fb'), os.path.join(tmp, 'co')
        subprocess.run([LLVM + 'llvm-objcopy', '--dump-section', '.hip_fatbin=' + fb, obj], check=True)
        subprocess.run([LLVM + 'clang-offload-bundler', '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--input=' + fb, '--output=' + co], check=True)
        txt = subprocess.run([LLVM + 'llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
    out, cur = [], None
    for line in txt.split('\n'):
        m = re.match(r'\s+-?\s*\.(\w+):\s+(.*)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == 'agpr_count':
            cur = {}
            out.append(cur)
        if cur is not None and k in ('agpr_count', 'vgpr_count', 'sgpr_count', 'private_segment_fixed_size', 'group_segment_fixed_size',
                                     'name', 'vgpr_spill_count', 'sgpr_spill_count', 'max_flat_workgroup_size'):
            cur[k] = v
    return out


def main():
    ks = kernels(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    names = subprocess.run(['c++filt'], input='\n'.join(k.get('name', '?') for k in ks), capture_output=True, text=True).stdout.strip().split('\n')
    print(f'{"kernel":<110s} {"VGPR":>4s} {"AGPR":>4s} {"SGPR":>4s} {"scratch":>7s} {"vspill":>6s} {"threads":>7s}')
    for k, n in zip(ks, names):
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'\(.*\)$', '', n)
        if flt and flt not in n:
            continue
        print(f'{n[:110]:<110s} {k.get("vgpr_count", "?"):>4s} {k.get("agpr_count", "?"):>4s} {k.get("sgpr_count", "?"):>4s} '
              f'{k.get("private_segment_fixed_size", "?"):>7s} {k.get("vgpr_spill_count", "?"):>6s} {k.get("max_flat_workgroup_size", "?"):>7s}')


if __name__ == '__main__':
    main()
