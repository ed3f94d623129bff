import re, sys
lines = open(sys.argv[1]).read().split('\n')
def regs(tok):
    # v[12:15] or v12 ; also a[..]
    out=[]
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b', tok):
        if m.group(1):
            out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3))+1)]
        else:
            out.append((m.group(4), int(m.group(5))))
    return out
lgkm=[]  # list of (dest regs set, line no) in issue order (LDS only; SMEM ignored -> conservative off)
vm=[]
viol=0
for n,l in enumerate(lines,1):
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
    op=t.split()[0]
    args=t[len(op):].split(';')[0]
    parts=[a.strip() for a in args.split(',')]
    if op=='s_waitcnt':
        m=re.search(r'lgkmcnt\((\d+)\)',t)
        if m:
            k=int(m.group(1)); lgkm=lgkm[len(lgkm)-k:] if k>0 else []
        m=re.search(r'vmcnt\((\d+)\)',t)
        if m:
            k=int(m.group(1)); vm=vm[len(vm)-k:] if k>0 else []
        continue
    if op.startswith('s_barrier') : continue
    # reads: for most VALU/MFMA: first operand is dst, rest are srcs; for stores all are srcs
    if op.startswith('ds_read') or op.startswith('ds_bpermute'):
        dst=set(regs(parts[0])); srcs=set(r for p in parts[1:] for r in regs(p))
    elif op.startswith('global_load') or op.startswith('buffer_load'):
        dst=set(regs(parts[0])); srcs=set(r for p in parts[1:] for r in regs(p))
    elif op.startswith('ds_write') or op.startswith('global_store') or op.startswith('buffer_store'):
        dst=set(); srcs=set(r for p in parts for r in regs(p))
    elif op.startswith('v_') :
        dst=set(regs(parts[0])); srcs=set(r for p in parts[1:] for r in regs(p))
        if op.startswith('v_mfma'): pass
    else:
        dst=set(); srcs=set()
    # check: any src or dst overlapping an outstanding load destination?
    for name,q in (('lgkm',lgkm),('vm',vm)):
        for d,ln in q:
            if d & srcs:
                print(f"RAW: line {n} `{t[:90]}` reads {sorted(d&srcs)[:4]} loaded at line {ln} ({name} outstanding {len(q)})"); viol+=1
            if d & dst:
                print(f"WAW: line {n} `{t[:90]}` overwrites {sorted(d&dst)[:4]} loaded at line {ln} ({name})"); viol+=1
    if op.startswith('ds_read') or op.startswith('ds_bpermute'): lgkm.append((dst,n))
    if op.startswith('global_load') or op.startswith('buffer_load'): vm.append((dst,n))
    if op.startswith('s_load') or op.startswith('s_buffer_load'): lgkm.append((set(),n))
print("violations (straight-line approximation, ignores branches):", viol)
