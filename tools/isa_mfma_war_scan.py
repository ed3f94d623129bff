import re, sys
lines=open(sys.argv[1]).read().split('\n')
def regs(tok):
    out=[]
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2))+1))
        else: out.append(int(m.group(3)))
    return set(out)
recent=[]  # (line, srcAB regs)
hits=[]
for n,l in enumerate(lines,1):
    t=l.strip()
    if not t or t[0] in ';.' or t.endswith(':'): continue
    op=t.split()[0]; parts=[a.strip() for a in t[len(op):].split(';')[0].split(',')]
    if op.startswith('v_mfma'):
        recent.append((n, regs(parts[1])|regs(parts[2])))
        recent=recent[-6:]
    elif op.startswith('ds_read') or op.startswith('global_load'):
        d=regs(parts[0])
        for ln,src in recent:
            if d & src and n-ln<=int(sys.argv[2]):
                hits.append((n, ln, op, sorted(d&src)[:4]))
    elif op.startswith('s_waitcnt') or op.startswith('s_nop'): pass
print(len(hits),"loads whose destination overlaps the A/B operands of an MFMA issued <=%s lines earlier"%sys.argv[2])
for h in hits[:12]: print("  load line %d (%s) overwrites operand regs %s of MFMA at line %d"%(h[0],h[2],h[3],h[1]))
