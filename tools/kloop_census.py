#!/usr/bin/env python3
"""Per-class instruction census of a kernel's K loop, from the disassembly (no GPU needed).

    python tools/kloop_census.py                       # every GEMM kernel of libdeepgemm_amd.so (one line each)
    python tools/kloop_census.py --dump duo_kernel<256,256,2,4,1,0,0,0,0,0,0,0,0>      # the loop's instructions
    python tools/kloop_census.py --asm file.s --label label_LoopBeginL                # a foreign disassembly (hipBLASLt)

The K loop = the innermost backward branch whose body holds the most MFMAs (the body between the branch target and the branch).
Classes: mfma, valu (everything else that starts with v_), ds (LDS reads / writes), vmem (buffer / global, LDS-DMA included), dma
(the `... lds` subset of vmem), salu, waitcnt, barrier, nop, branch.  "per K block" divides by (MFMAs in the body / MFMAs per K block of
one wave), the latter inferred from the tile (argument --mfma-per-kblock, default: MFMAs in the body if <= 64 else 32 or 64).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'deepgemm_amd', 'csrc', 'libdeepgemm_amd.so')
LLVM = '/opt/rocm/lib/llvm/bin'
CLASSES = ['mfma', 'valu', 'ds', 'vmem', 'dma', 'salu', 'waitcnt', 'barrier', 'nop', 'branch', 'total']


def classify(ins: str) -> str:
    op = ins.split()[0]
    if op.startswith('v_mfma') or op.startswith('v_smfma'):
        return 'mfma'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'ds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
        return 'vmem'
    if op == 's_waitcnt':
        return 'waitcnt'
    if op == 's_barrier':
        return 'barrier'
    if op in ('s_nop', 's_sleep'):
        return 'nop'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    return 'salu'


def parse(asm_text: str):
    """{symbol: [(address, instruction)]} from llvm-objdump -d output (labels inside a kernel do not split it when they are
    `label_*` -- Tensile's -- or anything not starting with _Z / Cijk / Custom)."""
    kernels, cur = {}, None
    for line in asm_text.splitlines():
        m = re.match(r'^([0-9a-f]+) <(\S+)>:', line)
        if m:
            name = m.group(2)
            if name.startswith(('_Z', 'Cijk', 'Custom')) or cur is None:
                cur = name
                kernels[cur] = []
            else:
                kernels[cur].append((int(m.group(1), 16), '.label ' + name))
            continue
        m = re.match(r'^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):', line)
        if m and cur is not None:
            kernels[cur].append((int(m.group(2), 16), m.group(1).strip()))
    return kernels


def find_loop(body, label=None):
    """(start index, end index) of the K loop in `body`."""
    addr_to_idx = {a: i for i, (a, ins) in enumerate(body) if not ins.startswith('.label')}
    if label:
        start = next(i for i, (a, ins) in enumerate(body) if ins == '.label ' + label)
        for i in range(start, len(body)):
            if body[i][1].startswith('s_cbranch') and label in body[i][1]:
                return start, i
        raise SystemExit('no branch back to ' + label)
    best = None
    for i, (addr, ins) in enumerate(body):
        m = re.match(r's_c?branch\w*\s+(\d+)', ins)
        if not m:
            continue
        off = int(m.group(1))
        if off < 0x8000:
            continue
        target = addr + 4 + (off - 0x10000) * 4
        j = addr_to_idx.get(target)
        if j is None or j >= i:
            continue
        n_mfma = sum(1 for _, x in body[j:i] if x.startswith('v_mfma'))
        # innermost = fewest instructions among the loops holding MFMAs; prefer the one with the most MFMAs per instruction
        if n_mfma and (best is None or (i - j) < (best[1] - best[0])):
            if best is None or n_mfma >= 16:
                best = (j, i)
    return best


def census(instrs):
    out = dict.fromkeys(CLASSES, 0)
    for ins in instrs:
        if ins.startswith('.label'):
            continue
        c = classify(ins)
        out[c] += 1
        out['total'] += 1
        if c == 'vmem' and ins.rstrip().endswith(' lds'):
            out['dma'] += 1
    return out


def pretty(name):
    m = re.match(r'_ZN2dg\d+(\w+?)(?:I(.*?)EEv|Ev)', name)
    if not m:
        return name[:100]
    return m.group(1) + ('<' + ','.join(re.findall(r'L[ib](\d+)E', m.group(2))) + '>' if m.group(2) else '')


def library_asm():
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, 'lib.so')
        with open(LIB, 'rb') as src, open(local, 'wb') as dst:
            dst.write(src.read())
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        obj = os.path.join(tmp, [f for f in os.listdir(tmp) if 'amdgcn' in f][0])
        return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', obj], check=True, capture_output=True, text=True).stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--asm', help='a disassembly file instead of the library')
    ap.add_argument('--label', help='loop label (Tensile kernels)')
    ap.add_argument('--dump', help='print the loop of the kernel whose pretty name contains this')
    ap.add_argument('--mfma-per-kblock', type=int, default=0)
    ap.add_argument('--only', default='gemm', help='substring filter on the symbol')
    args = ap.parse_args()
    text = open(args.asm).read() if args.asm else library_asm()
    kernels = parse(text)
    print(f"{'kernel':64s} " + ' '.join(f'{c:>7s}' for c in CLASSES) + '   (per K block of one wave; K blocks per loop body)')
    for name, body in kernels.items():
        if args.only not in name and not args.asm:
            continue
        loop = find_loop(body, args.label)
        if loop is None:
            continue
        instrs = [ins for _, ins in body[loop[0]:loop[1] + 1]]
        c = census(instrs)
        per = args.mfma_per_kblock or (c['mfma'] if c['mfma'] <= 64 else 64 if c['mfma'] % 64 == 0 and 'quad' in name else 32)
        blocks = max(c['mfma'] / per, 1e-9)
        p = pretty(name)
        print(f'{p[:64]:64s} ' + ' '.join(f'{c[k] / blocks:7.1f}' for k in CLASSES) + f'   ({blocks:g})')
        if args.dump and args.dump in p:
            for ins in instrs:
                print('    ' + ins)


if __name__ == '__main__':
    main()
