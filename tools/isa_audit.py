#!/usr/bin/env python3
"""ISA audit of the loops of every kernel in gfx950 assembly files (hipcc -save-temps): instruction mix, the share of v_cmp / v_cndmask
(register arrays indexed by a run-time value), and vector-memory loads whose s_waitcnt vmcnt(0) follows within a few instructions (a
'prefetch' the compiler serialised).  Round 5 found fiber_conv_bwd_kernel (45 % cmp / cndmask) and lift_encode_bwd (four serialised
loads per node in the bf16 build) this way.   usage: tools/isa_audit.py file.s [file.s ...]"""
import collections
import re
import sys


def kernels(path):
    s = open(path).read()
    for m in re.finditer(r'^(_Z\w+):.*\n', s, re.M):
        e = s.find('s_endpgm', m.end())
        if e > 0 and '.amdhsa_kernel' not in s[m.end():e]:
            yield m.group(1), [l.strip() for l in s[m.end():e].split('\n')]


def audit(path):
    for name, body in kernels(path):
        body = [l.split(';')[0].strip() for l in body]
        body = [l for l in body if l]
        labels = {l[:-1]: i for i, l in enumerate(body) if re.match(r'^\.LBB\w+:$', l)}
        for i, l in enumerate(body):
            m = re.match(r's_c?branch\S*\s+(\.LBB\w+)', l)
            if not (m and m.group(1) in labels and labels[m.group(1)] < i):
                continue
            loop = [x.split()[0] for x in body[labels[m.group(1)]:i] if not x.endswith(':') and not x.startswith('.')]
            if len(loop) < 24:
                continue
            c = collections.Counter(loop)
            sel = sum(v for k, v in c.items() if k.startswith(('v_cmp', 'v_cndmask')))
            serial = 0
            seg = body[labels[m.group(1)]:i]
            for j, x in enumerate(seg):
                if x.startswith(('global_load', 'buffer_load', 'flat_load')):
                    for y in seg[j + 1:j + 4]:
                        if y.startswith('s_waitcnt vmcnt(0)'):
                            serial += 1
                            break
                        if y.startswith(('global_load', 'buffer_load')):
                            break
            flag = ('  <-- select-heavy' if sel > 0.2 * len(loop) else '') + ('  <-- serialised loads' if serial >= 2 else '')
            print(f"{name[:70]:70s} loop {len(loop):4d} sel {sel:3d} serial-loads {serial} movs {c['v_mov_b32_e32']:3d} top {c.most_common(4)}{flag}")


if __name__ == '__main__':
    for p in sys.argv[1:]:
        print('==', p)
        audit(p)
