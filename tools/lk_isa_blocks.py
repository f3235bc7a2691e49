#!/usr/bin/env python3
"""Static view of lk_levels_kernel<4>'s ISA: VALU / LDS / scratch instruction counts per basic block, inline-asm blocks apart.
usage: hipcc ... --cuda-device-only -S ofps_amd/csrc/lk.hip -o /tmp/lk.s ; lk_isa_blocks.py /tmp/lk.s [min_valu]"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
minv = int(sys.argv[2]) if len(sys.argv) > 2 else 12
start = [i for i, l in enumerate(lines) if l.startswith('_ZN4ofps16lk_levels_kernelILi4EEEvNS_12LkLevelsArgsE:')][0]
fe = [i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end')][0]
blocks = []; cur = ['entry', []]
for l in lines[start:fe]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]
    else:
        cur[1].append(l.strip())
blocks.append(cur)
for name, ls in blocks:
    inapp = False; v_in = v_out = lds = sc = bar = 0; br = []
    for l in ls:
        if l.startswith(';APP'): inapp = True; continue
        if l.startswith(';NO_APP'): inapp = False; continue
        if l.startswith('v_'):
            if inapp: v_in += 1
            else: v_out += 1
        if l.startswith('ds_'): lds += 1
        if l.startswith('scratch_') or l.startswith('buffer_'): sc += 1
        if l.startswith('s_barrier'): bar += 1
        if l.startswith('s_cbranch') or l.startswith('s_branch'): br.append(l.split()[-1])
    if v_in + v_out >= minv or sc or bar:
        print(f"{name:12s} valu_asm {v_in:5d} valu_c {v_out:4d} lds {lds:4d} scratch {sc:3d} barrier {bar} -> {' '.join(br)}")
