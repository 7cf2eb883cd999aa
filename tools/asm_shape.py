#!/usr/bin/env python3
"""Run-length picture of a kernel's main basic block in a hipcc -S listing: A = fp32 MFMA, B = fp64 MFMA, e = transcendental,
. = other VALU, d = LDS, g = global memory, w = s_waitcnt, n = s_nop.  usage: asm_shape.py listing.s mangled_kernel_name"""
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
i = s.find(name + ':'); j = s.find('.end_amdhsa_kernel', i)
seq = []
for l in s[i:j].split('\n'):
    t = l.strip().split(' ')[0]
    if t.startswith('v_mfma_f32'): seq.append('A')
    elif t.startswith('v_mfma_f64'): seq.append('B')
    elif t.startswith(('v_exp', 'v_log', 'v_rcp')): seq.append('e')
    elif t.startswith('v_'): seq.append('.')
    elif t.startswith('ds_'): seq.append('d')
    elif t.startswith(('global_', 'buffer_')): seq.append('g')
    elif t.startswith('s_waitcnt'): seq.append('w')
    elif t.startswith('s_nop'): seq.append('n')
    elif t.startswith(('s_cbranch', 's_branch')): seq.append('|')
txt = ''.join(seq)
print([(x.count('A'), x.count('B'), len(x)) for x in txt.split('|') if x.count('A') + x.count('B') > 0])
for sg in sorted(txt.split('|'), key=lambda x: -(x.count('A') + x.count('B')))[:1]:
    out = []; prev = None; c = 0
    for ch in sg:
        if ch == prev: c += 1
        else:
            if prev: out.append(f"{prev}{c}")
            prev = ch; c = 1
    out.append(f"{prev}{c}")
    print(' '.join(out))
