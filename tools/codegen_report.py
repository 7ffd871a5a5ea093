#!/usr/bin/env python3
"""Static look at what hipcc made of every GEMM kernel in libdeepgemm_amd.so: VGPR / spill counts from the code-object notes
and, from the disassembly, the number of scratch (spill) instructions between the first and the last MFMA of the kernel --
the K loop.  A spill inside the K loop costs far more than its instruction: scratch traffic counts towards vmcnt and
tightens every counted wait.  Used by tests/test_codegen.py.   python tools/codegen_report.py [--json]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'deepgemm_amd', 'csrc', 'libdeepgemm_amd.so')
LLVM = '/opt/rocm/lib/llvm/bin'


_REG = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
_VMEM = ('buffer_load', 'buffer_store', 'buffer_atomic', 'global_load', 'global_store', 'global_atomic', 'scratch_load',
         'scratch_store', 'flat_load', 'flat_store')


def _vregs(text):
    regs = set()
    for m in _REG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def landing_hazards(body):
    """The "asm load" rule, checked on the instruction stream: a vector-memory load that lands in VGPRs makes them valid only
    after an s_waitcnt whose vmcnt leaves no more operations outstanding than were issued after it (vector-memory operations
    of a wave retire in order).  Until then no instruction may touch those registers (hipcc treats the destinations of an
    inline-asm load as ordinary values and has copied them early once, HISTORY.md "A latent race"), and no branch may be taken:
    the rule the kernels follow is "an asm load reaches its wait in straight-line code".  Returns the violations found in the
    linear instruction sequence `body`: (index, kind, instruction) with kind 'touch' or 'branch'."""
    outstanding = []            # per vector-memory operation in issue order: set of landing VGPRs (empty for stores / LDS-DMA)
    bad = []
    for idx, ins in enumerate(body):
        op = ins.split()[0] if ins.split() else ''
        if op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', ins)
            if m:
                keep = int(m.group(1))
                if len(outstanding) > keep:
                    outstanding = outstanding[len(outstanding) - keep:] if keep else []
            continue
        in_flight = set().union(*outstanding) if outstanding else set()
        if op.startswith(_VMEM):
            operands = ins[len(op):]
            first = operands.split(',')[0]
            # (scratch reloads are the compiler's own, waited for by its own counters: hipcc does emit back-to-back reloads into
            #  one register, e.g. `scratch_load_dwordx2 v[56:57]` + `scratch_load_dword v56`; the rule is about INLINE-ASM loads)
            is_load = 'load' in op and not ins.rstrip().endswith(' lds') and not op.startswith('scratch_')
            dst = _vregs(first) if is_load else set()
            rest = _vregs(operands) - dst if is_load else _vregs(operands)
            if in_flight & rest or (in_flight & dst):
                bad.append((idx, 'touch', ins))
            outstanding.append(dst)
            continue
        if in_flight:
            if op.startswith(('s_cbranch', 's_branch', 's_setpc', 's_endpgm')):
                # only loads issued by inline asm matter for the branch rule, but the disassembly cannot tell them apart:
                # report it, the caller decides
                bad.append((idx, 'branch', ins))
                outstanding = [set() for _ in outstanding]      # (a new block: stop tracking, the touch rule restarts)
                continue
            if in_flight & _vregs(ins[len(op):]):
                bad.append((idx, 'touch', ins))
    return bad


_SREG = re.compile(r'\bs(\d+)\b|\bs\[(\d+):(\d+)\]')


def _sregs(text):
    regs = set()
    for m in _SREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def sgpr_vmem_hazards(body, need=5):
    """gfx950 data hazard "VALU writes an SGPR -> a vector-memory instruction reads it (descriptor or scalar offset)": 5 wait states.
    hipcc pads its own code, but NOTHING pads an inline-asm string: round 3 found `v_readlane_b32 s6, v213, 37` (an SGPR reloaded from a
    spill lane) directly in front of an asm `buffer_load_dwordx4 ..., s6 offen` -- the load went out with the stale offset and whole wave
    tiles came out wrong, run-dependent.  The asm scale-load blocks now open with `s_nop 4`; this check keeps it that way.  Returns
    (index, writer instruction, reader instruction) for every VMEM instruction with fewer than `need` wait states behind a VALU write of
    one of its SGPRs (v_readlane / v_readfirstlane / any v_* with an SGPR destination) in the linear stream."""
    bad = []
    for idx, ins in enumerate(body):
        op = ins.split()[0] if ins.split() else ''
        if not op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')):
            continue
        used = _sregs(ins[len(op):])
        if not used:
            continue
        states = 0
        for back in range(idx - 1, max(idx - 1 - need, -1), -1):
            prev = body[back]
            pop = prev.split()[0] if prev.split() else ''
            if pop.startswith('v_'):
                first = prev[len(pop):].split(',')[0]
                if pop.startswith(('v_readlane', 'v_readfirstlane')) or re.match(r'\s*s(\d+|\[)', first):
                    if used & _sregs(first):
                        bad.append((idx, prev, ins))
                        break
            if pop.startswith(('s_cbranch', 's_branch', 's_setpc', 's_endpgm')):
                break                                   # (another block: not a linear predecessor)
            states += 1 + (int(prev.split()[1]) if pop == 's_nop' and len(prev.split()) > 1 and prev.split()[1].isdigit() else 0)
            if states >= need:
                break
    return bad


def report():
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, 'lib.so')
        with open(LIB, 'rb') as src, open(local, 'wb') as dst:
            dst.write(src.read())
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        # one device code object per translation unit of the (sharded) build: deepgemm_amd/build.py
        device = sorted(f for f in os.listdir(tmp) if 'amdgcn' in f)
        assert device, 'no device code object found in the library'
        notes, asm = '', ''
        for f in device:
            obj = os.path.join(tmp, f)
            notes += subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', obj], check=True, capture_output=True, text=True).stdout
            asm += subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', obj], check=True, capture_output=True, text=True).stdout
    meta = {}
    name = None
    for line in notes.splitlines():
        m = re.search(r'\.name:\s+(\S+)', line)
        if m:
            name = m.group(1)
            meta[name] = {}
        for key in ('vgpr_count', 'vgpr_spill_count', 'sgpr_spill_count'):
            m = re.search(r'\.%s:\s+(\d+)' % key, line)
            if m and name:
                meta[name][key] = int(m.group(1))
    kernels = {}
    cur = None
    for line in asm.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if cur is not None and line.strip():
            kernels[cur].append(line.split('//')[0].strip())
    out = []
    for name, body in kernels.items():
        if name not in meta or 'gemm' not in name:
            continue
        mfma = [i for i, ins in enumerate(body) if ins.startswith('v_mfma')]
        # the K loop: from the first MFMA to the first backward branch behind it (kernels with a K-tail stage have more MFMAs after
        # the loop; spills there cost once per tile, not once per K block); no backward branch found: up to the last MFMA
        end = mfma[-1] if mfma else 0
        if mfma:
            for i in range(mfma[0], mfma[-1]):
                m_br = re.match(r's_c?branch\w*\s+(\d+)', body[i])
                if m_br and int(m_br.group(1)) >= 0x8000:
                    end = max(i, mfma[min(len(mfma) - 1, 15)])
                    break
        loop = body[mfma[0]:end + 1] if mfma else []
        # readable name without a demangler: _ZN2dg22dg_fp8_gemm_duo_kernelILi256ELi256ELi2ELi4ELi0EEEvNS_10GemmParamsE
        m = re.match(r'_ZN2dg\d+(\w+?)(?:I(.*?)EEv|Ev)', name)
        pretty = name if not m else m.group(1) + ('<' + ','.join(re.findall(r'L[ib](\d+)E', m.group(2))) + '>' if m.group(2) else '')
        out.append({'kernel': pretty, 'symbol': name, **meta[name],
                    'mfma_range_instructions': len(loop),
                    'scratch_in_mfma_range': sum(1 for ins in loop if ins.startswith('scratch_')),
                    'lane_ops_in_mfma_range': sum(1 for ins in loop if ins.startswith(('v_readlane', 'v_writelane'))),
                    # hipcc's waterfall loop around a buffer operation whose descriptor is not provably wave-uniform (v_readfirstlane x 4,
                    # two v_cmp_eq_u64, exec save / restore, a branch per LDS-DMA piece): round 4 found them in every stream / pipe kernel.
                    # Counted from 200 instructions ahead of the first MFMA (the stage issue code sits in front of it) to the loop's end.
                    'waterfalls_at_k_loop': sum(1 for ins in body[max(mfma[0] - 200, 0):end + 1] if ins.startswith('v_cmp_eq_u64')) // 2 if mfma else 0,
                    'landing_touches': [h for h in landing_hazards(body) if h[1] == 'touch'],
                    'sgpr_vmem_hazards': sgpr_vmem_hazards(body),
                    'landing_branches_in_mfma_range': [h for h in landing_hazards(loop) if h[1] == 'branch']})
    return out


if __name__ == '__main__':
    rows = report()
    if '--json' in sys.argv:
        print(json.dumps(rows))
    else:
        for r in sorted(rows, key=lambda r: r['kernel']):
            print(f"{r['kernel'][:78]:78s} vgpr {r.get('vgpr_count', -1):3d} spill {r.get('vgpr_spill_count', -1):3d} "
                  f"range {r['mfma_range_instructions']:5d} scratch-in-range {r['scratch_in_mfma_range']:3d} "
                  f"lane-ops-in-range {r['lane_ops_in_mfma_range']:3d} waterfalls {r['waterfalls_at_k_loop']:2d} landing-touches {len(r['landing_touches']):2d} sgpr->vmem hazards {len(r['sgpr_vmem_hazards']):2d} branches-in-range-with-loads-in-flight {len(r['landing_branches_in_mfma_range']):2d}")
