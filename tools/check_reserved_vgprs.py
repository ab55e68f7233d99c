"""The A-operand sets of csrc/stgcn_tconv3.hip live in v224..v255, which the compiler must never allocate (the kernels
carry amdgpu_num_vgpr(224); the attribute is a target, not a guarantee: under pressure the allocator goes past it
rather than spill -- found in round 3).  This compiles the source to assembly and reports every instruction outside the
inline-assembly blocks that names one of those registers.
    python tools/check_reserved_vgprs.py [source.hip]   -> exit status 1 if any"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIRST = 224


def offenders(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '--offload-arch=gfx950',
                        '-I' + os.path.join(ROOT, 'pose2room_amd', 'csrc'), '-S', '--cuda-device-only', '-o', out, src],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split('\n')
    bad, inasm, kernel = [], False, '?'
    for n, line in enumerate(text, 1):
        s = line.strip()
        if s.endswith(':') and s.startswith('_Z'):
            kernel = s[:60]
        if 'ASMSTART' in s:
            inasm = True
            continue
        if 'ASMEND' in s:
            inasm = False
            continue
        if inasm or not s or s[0] in ';.':
            continue
        regs = [int(m) for m in re.findall(r'\bv(\d+)\b', s)]
        regs += [r for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', s) for r in range(int(a), int(b) + 1)]
        if any(r >= FIRST for r in regs):
            bad.append((kernel, n, s))
    return bad


if __name__ == '__main__':
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'pose2room_amd', 'csrc', 'stgcn_tconv3.hip')
    bad = offenders(src)
    for k, n, s in bad[:20]:
        print(f'{k} line {n}: {s}')
    print(f'{len(bad)} compiler-generated instructions touch v{FIRST}+')
    sys.exit(1 if bad else 0)
