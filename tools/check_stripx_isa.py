#!/usr/bin/env python3
"""Checks the ISA of csrc/conv_f16s_stripx.hip for the two assumptions its hand-placed waits rest on:
 1. no instruction touches the destination of an inline-assembly ds_read between that read and the next s_waitcnt lgkmcnt(0)
    (the compiler does not know the value is still in flight: a copy or a spill there would read stale registers);
 2. inside the tile loop (from the barrier behind the counted waits to the next one) there is no vmcnt(0) — the form every
    compiler-inserted vmcnt wait takes in this kernel — and no scratch access.
Usage: python tools/check_stripx_isa.py [file.s]   (default: compiles the source for gfx950 into a temporary file)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'ood-gan-inversion_amd', 'csrc', 'conv_f16s_stripx.hip')
SRC8 = os.path.join(ROOT, 'ood-gan-inversion_amd', 'csrc', 'experimental', 'conv_f16s_stripx8.hip')      # the eight-wave forward instance: same inline-assembly discipline


def regs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check(path):
    text = open(path).read()
    bad = 0
    for fn in re.finditer(r'^(_Z\S*stripx8?_(?:fwd_)?kernel\S*):[^\n]*\n(.*?)s_endpgm\n', text, re.S | re.M):
        name, body = fn.group(1), fn.group(2).split('\n')
        pending, in_asm, n_reads = set(), False, 0
        for ln, line in enumerate(body):
            code = line.split(';')[0].strip()
            if 'ASMSTART' in line:
                in_asm = True
                continue
            if 'ASMEND' in line:
                in_asm = False
                continue
            if not code or code.endswith(':'):
                continue
            if code.startswith('s_waitcnt') and 'lgkmcnt(0)' in code:
                pending.clear()
                continue
            if in_asm and code.startswith('ds_read'):
                dst = code.split()[1].rstrip(',')
                pending |= regs(dst)
                n_reads += 1
                continue
            touched = regs(code) & pending
            if touched:
                print(f'{name}: line {ln}: `{code}` touches in-flight LDS destination(s) v{sorted(touched)}')
                bad += 1
        bars = [i for i, l in enumerate(body) if l.strip().startswith('s_barrier')]
        # the tile loop: from the barrier that follows the counted waits to the barrier after the loop
        loop_bar = [i for i in bars if any(re.search(r'vmcnt\(([1-9]\d*)\)', body[j]) for j in range(max(0, i - 12), i))]
        if loop_bar:
            start = loop_bar[0]
            end = next(i for i in bars if i > start)
            # a role loop of the eight-wave kernel ends at its backward branch, not at a barrier (the next barrier belongs to the other role)
            labels = {l.split(';')[0].strip()[:-1]: i for i, l in enumerate(body) if l.split(';')[0].strip().endswith(':')}
            for i in range(start + 1, end):
                c = body[i].split(';')[0].strip()
                if c.startswith('s_branch') and labels.get(c.split()[1], 1 << 30) < i:
                    end = i + 1
                    break
            for i in range(start + 1, end):
                c = body[i].split(';')[0].strip()
                if c.startswith('scratch_'):
                    print(f'{name}: line {i}: scratch access `{c}` inside the tile loop')
                    bad += 1
                # the explicit counted waits are vmcnt(N > 0); a wait the compiler adds for a register load or an LDS access that
                # may alias a pending LDS-DMA comes out as vmcnt(0) (its model treats the DMA as a pending FLAT access)
                if c.startswith('s_waitcnt') and 'vmcnt(0)' in c and i < end - 3:      # (the wait of the loop exit sits right before `end`)
                    print(f'{name}: line {i}: unexpected `{c}` inside the tile loop')
                    bad += 1
        print(f'{name[:60]}...: {n_reads} inline LDS reads checked, loop lines {loop_bar[:1]}')
    return bad


if __name__ == '__main__':
    if len(sys.argv) > 1:
        sys.exit(1 if check(sys.argv[1]) else 0)
    with tempfile.TemporaryDirectory() as d:
        bad = 0
        for src in (SRC, SRC8):
            out = os.path.join(d, os.path.basename(src) + '.s')
            subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Xclang', '-target-feature', '-Xclang',
                                   '-packed-fp32-ops', '-S', '--cuda-device-only', '-o', out, src], stderr=subprocess.DEVNULL)
            bad += check(out)
        sys.exit(1 if bad else 0)
