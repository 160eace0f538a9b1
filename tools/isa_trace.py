#!/usr/bin/env python3
"""Print a one-letter-per-instruction trace of a kernel from a hipcc -save-temps .s file.
M=mfma v=valu E=v_exp L=lds G=global w=waitcnt B=barrier s=salu j=branch
usage: isa_trace.py file.s <substring of the mangled kernel name>"""
import sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and pat in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
body = [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith(('.', ';', '//'))]
def cls(l):
    op = l.split()[0]
    if op.endswith(':'): return '\n' + op + ' '
    if op.startswith('v_mfma'): return 'M'
    if op.startswith('ds_'): return 'L'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'G'
    if op.startswith('s_barrier'): return 'B'
    if op.startswith('s_waitcnt'): return 'w'
    if op.startswith(('s_cbranch', 's_branch')): return 'j'
    if op.startswith('s_'): return 's'
    if op.startswith('v_exp'): return 'E'
    if op.startswith('v_'): return 'v'
    return '?'
print(lines[start].split(':')[0], len(body), 'instructions')
print(''.join(cls(l) for l in body))
