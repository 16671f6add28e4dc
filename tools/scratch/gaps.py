"""Instruction classes between consecutive MFMAs of a loop: python gaps.py FILE.s FIRST_LINE LAST_LINE"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')[int(sys.argv[2]) - 1:int(sys.argv[3]) - 1]
gap = []
n = 0
def cls(op):
    if op.startswith('v_exp'): return 'E'
    if op.startswith('v_pk_'): return 'P'
    if op.startswith('v_accvgpr'): return 'a'
    if op.startswith('v_'): return 'v'
    if op.startswith('ds_read'): return 'L'
    if op.startswith('global_load_lds'): return 'D'
    if op.startswith('global_store'): return 'G'
    if op == 's_nop': return 'n'
    if op == 's_waitcnt': return 'w'
    if op == 's_barrier': return 'B'
    if op.startswith('s_'): return 's'
    return '?'
out = []
for l in lines:
    m = re.match(r'\s+([a-z_0-9]+)', l)
    if not m: continue
    op = m.group(1)
    if op.startswith('v_mfma'):
        out.append(''.join(gap)); gap = []; n += 1
    else:
        gap.append(cls(op))
out.append(''.join(gap))
for i in range(0, len(out), 4):
    print(' | '.join('%-22s' % g for g in out[i:i + 4]))
print(n, 'MFMAs; E exp (8 cyc) v VALU P packed L ds_read D lds-dma G store n s_nop w waitcnt s SALU B barrier')
