"""Static scan of the device ISA for SERIALIZED global round trips: a `global_load` / `buffer_load` whose result is waited for
(`s_waitcnt vmcnt(0)`) before any other load is issued.  Every such pair in the straight-line part of a kernel (its prologue, or a
divergent `if (valid) x = p[i]` block) is a full L2 / HBM latency in front of the kernel's first useful instruction; in a frame of
5-15 us launches that is 5-10 % of a launch each (found with this: aff_score_kernel's c_j / tau_j loads, round 4).
Usage: hipcc ... --cuda-device-only -S file.hip -o file.s ; python tools/isa_waits.py file.s [kernel-name-substring]
       python tools/isa_waits.py cutie_amd/csrc/affinity.o [kernel-name-substring]      (a built object: disassembled, no recompile)"""
import re, sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r'^(_Z\w+|\w+):\s*; @', line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
        elif name is not None:
            if line.startswith('.Lfunc_end'):
                yield name, body
                name, body = None, []
            else:
                body.append(line.rstrip())
    if name:
        yield name, body


LLVM = '/opt/rocm/lib/llvm/bin/'


def kernels_of_object(obj):
    """The same (name, body) pairs from a built object file (cutie_amd/csrc/*.o after `make`): the gfx950 code object is taken out of the
    fat binary and disassembled (llvm-objdump --symbolize-operands), labels and branch operands rewritten to the `.LBB` form of `hipcc -S`
    so that scan() / wait_groups() see the same text.  No GPU and no recompile: this is what tests/test_isa_guard_cpu.py runs."""
    import os, subprocess, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        fb, co = os.path.join(tmp, 'fb'), os.path.join(tmp, 'co')
        subprocess.run([LLVM + 'llvm-objcopy', '--dump-section', '.hip_fatbin=' + fb, obj], check=True)
        subprocess.run([LLVM + 'clang-offload-bundler', '--unbundle', '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--input=' + fb, '--output=' + co], check=True)
        txt = subprocess.run([LLVM + 'llvm-objdump', '-d', '--symbolize-operands', co], capture_output=True, text=True, check=True).stdout
    name, body = None, []
    for line in txt.split('\n'):
        m = re.match(r'^[0-9a-f]+ <([^>]+)>:', line)
        if m:
            lab = m.group(1)
            if re.fullmatch(r'L\d+', lab):
                if name is not None:
                    body.append('.LBB_' + lab[1:] + ':')
                continue
            if name is not None:
                yield name, body
            name, body = lab, []
            continue
        if name is None or not line.startswith('\t'):
            continue
        ins = line.split('//')[0].rstrip()
        ins = re.sub(r'\bL(\d+)\b', r'.LBB_\1', ins) if re.match(r'\s*s_c?branch', ins) else ins
        body.append(ins)
    if name is not None:
        yield name, body


def scan(body):
    """-> list of (line index, load text, instructions between load and wait, inside a loop?)"""
    labels = {}
    for n, l in enumerate(body):
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            labels[m.group(1)] = n
    loops = []                                                    # (start, end) of backward branches
    for n, l in enumerate(body):
        m = re.search(r's_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)', l)
        if m:
            t = labels.get(m.group(1) or m.group(2))
            if t is not None and t < n:
                loops.append((t, n))
    out = []
    pending = None                                                # (index, text, other instructions since)
    outstanding = 0
    for n, l in enumerate(body):
        s = l.strip()
        if not s or s.startswith(';') or s.endswith(':') or s.startswith('.'):
            continue
        if re.match(r'(global_load|buffer_load|flat_load|scratch_load)', s):
            if 'lds' in s.split()[-1:]:
                pass
            outstanding += 1
            pending = (n, s, 0) if outstanding == 1 else None
            continue
        m = re.match(r's_waitcnt.*vmcnt\((\d+)\)', s)
        if m and int(m.group(1)) == 0:
            if pending is not None and outstanding == 1:
                out.append((pending[0], pending[1], pending[2], any(a <= pending[0] <= b for a, b in loops)))
            outstanding = 0
            pending = None
            continue
        if m:
            outstanding = min(outstanding, int(m.group(1)))
            continue
        if pending is not None:
            pending = (pending[0], pending[1], pending[2] + 1)
    return out


def wait_groups(body):
    """(before the first loop, whole kernel): number of vmcnt waits -- of any count -- with at least one load issued since the previous
    one.  A kernel whose loads are all in flight together has 1-3; `load, wait, load, wait ...` shows up as one group per load."""
    labels = {}
    for n, l in enumerate(body):
        m = re.match(r'^(\.LBB\w+):', l)
        if m:
            labels[m.group(1)] = n
    firstloop = len(body)
    for n, l in enumerate(body):
        m = re.search(r's_c?branch\w*\s+(\.LBB\w+)', l)
        if m:
            t = labels.get(m.group(1))
            if t is not None and t < n:
                firstloop = min(firstloop, t)
    pre = tot = since = 0
    for n, l in enumerate(body):
        s = l.strip()
        if re.match(r'(global_load|buffer_load|flat_load)', s):
            since += 1
        elif re.match(r's_waitcnt.*vmcnt\(', s):
            if since:
                tot += 1
                pre += n < firstloop
            since = 0
    return pre, tot


if __name__ == '__main__':
    if sys.argv[1] == '--groups':                              # python tools/isa_waits.py --groups file.s [name substrings ...]
        for name, body in kernels(sys.argv[2]):
            if len(sys.argv) > 3 and not any(s in name for s in sys.argv[3:]):
                continue
            pre, tot = wait_groups(body)
            print(f'{pre:3d} wait groups before the first loop, {tot:3d} in all   {name[:110]}')
        sys.exit(0)
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ''
    import subprocess
    for name, body in (kernels_of_object(path) if path.endswith('.o') else kernels(path)):
        if sub and sub not in name:
            continue
        hits = scan(body)
        if not hits:
            continue
        try:
            dem = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
        except Exception:
            dem = name
        pro = [h for h in hits if not h[3]]
        print(f'{dem[:110]}: {len(pro)} lone load+wait outside loops, {len(hits) - len(pro)} inside loops')
        for n, s, gap, inloop in hits[:12]:
            print(f'    line {n:5d} {"loop" if inloop else "    "} {gap:3d} instr before the wait: {s}')
