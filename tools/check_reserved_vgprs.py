"""Kernels that keep operands in vector registers BY NAME inside inline assembly (csrc/stgcn_tconv3.hip: the A-operand
sets v224..v255) rely on the compiler never allocating those registers.  The kernels carry amdgpu_num_vgpr(224), but
the attribute is a target, not a guarantee: under pressure the allocator goes past it rather than spill (found in
round 3: a variant silently overwrote the operands of the next tile).

This tool compiles a source to assembly WITH THE FLAGS OF THE BUILD and reports every compiler-generated instruction
(i.e. outside the ASMSTART/ASMEND blocks) that names one of the registers the source's own inline assembly names.
The Makefile runs it over every *.hip before it links libp2r_hip.so (target `.reserved_vgprs.ok`), so a toolchain or
flag change that breaks the reservation fails the BUILD, not a test somebody may not run.

    python tools/check_reserved_vgprs.py [--hipcc HIPCC] [--flags "..."] [source.hip ...]   -> exit status 1 if any
"""
import argparse
import os
import re
import shlex
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pose2room_amd', 'csrc')
DEFAULT_HIPCC = '/opt/rocm/bin/hipcc'
DEFAULT_FLAGS = '-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function'


def named_registers(src):
    """VGPR numbers that the source names literally in string literals (inline assembly text and clobber lists)."""
    text = open(src).read()
    text = re.sub(r'//[^\n]*', '', text)
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    regs = set()
    for lit in re.findall(r'"((?:[^"\\]|\\.)*)"', text):
        regs.update(int(m) for m in re.findall(r'\bv(\d+)\b', lit))
        for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', lit):
            regs.update(range(int(a), int(b) + 1))
    return regs


def offenders(src, hipcc=DEFAULT_HIPCC, flags=DEFAULT_FLAGS, reserved=None):
    reserved = named_registers(src) if reserved is None else set(reserved)
    if not reserved:
        return []
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        cmd = [hipcc] + shlex.split(flags) + ['-I' + CSRC, '-S', '--cuda-device-only', '-o', out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
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
        if any(r in reserved for r in regs):
            bad.append((kernel, n, s))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hipcc', default=DEFAULT_HIPCC)
    ap.add_argument('--flags', default=DEFAULT_FLAGS)
    ap.add_argument('sources', nargs='*')
    args = ap.parse_args()
    sources = args.sources or [os.path.join(CSRC, 'stgcn_tconv3.hip')]
    failed = 0
    for src in sources:
        reserved = named_registers(src)
        if not reserved:
            continue
        bad = offenders(src, args.hipcc, args.flags, reserved)
        for k, n, s in bad[:20]:
            print(f'{os.path.basename(src)}: {k} line {n}: {s}')
        print(f'{os.path.basename(src)}: names v{min(reserved)}..v{max(reserved)} in inline assembly; '
              f'{len(bad)} compiler-generated instructions touch them')
        failed += bool(bad)
    sys.exit(1 if failed else 0)


if __name__ == '__main__':
    main()
