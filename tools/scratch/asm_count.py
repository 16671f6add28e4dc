import re,collections,sys
lines=open(sys.argv[1]).read().split('\n')
a,b=int(sys.argv[2]),int(sys.argv[3])
c=collections.Counter()
for l in lines[a-1:b-1]:
    m=re.match(r'\s+([a-z_0-9]+)',l)
    if not m: continue
    op=m.group(1)
    if op.startswith('v_mfma'): k='mfma'
    elif op.startswith('v_'): k=op.replace('_e32','').replace('_e64','')
    elif op.startswith(('ds_','global','scratch','buffer')): k=op
    elif op in('s_waitcnt','s_barrier','s_nop'): k=op
    elif op.startswith('s_'): k='salu'
    else: k=op
    c[k]+=1
tot=sum(v for k,v in c.items() if k.startswith('v_'))
print('VALU',tot,dict(sorted(c.items(),key=lambda x:-x[1])))
