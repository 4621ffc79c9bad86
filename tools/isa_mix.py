#!/usr/bin/env python3
"""Dev helper: static instruction mix per basic block of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: python tools/isa_mix.py /tmp/mid.s k_mid_layer_fwdILi32ELb0ELb0ELb0ELi128E"""
import collections, re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\S*%s\S*):' % re.escape(sys.argv[2]), s, re.M)
body = s[m.start():s.index('s_endpgm', m.start())]
blocks, cur = [], ('entry', [])
for l in body.split('\n'):
    l = l.strip()
    if not l or l.startswith(';'):
        continue
    if l.startswith('.LBB') and l.split()[0].endswith(':'):
        blocks.append(cur); cur = (l.split(':')[0], [])
    elif l.startswith('.') or l.endswith(':'):
        continue
    else:
        cur[1].append(l)
blocks.append(cur)
tot = collections.Counter()
for name, b in blocks:
    c = collections.Counter()
    for l in b:
        op = l.split()[0]
        k = ('mfma' if op.startswith('v_mfma') else 'readlane' if op.startswith(('v_readlane', 'v_readfirstlane')) else
             'valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else
             'smem' if op.startswith('s_load') else 'wait' if op.startswith(('s_waitcnt', 's_nop', 's_barrier')) else 'branch' if op.startswith(('s_cbranch', 's_branch')) else 'salu')
        c[k] += 1
    tot.update(c)
    tail = b[-1] if b else ''
    print(name.ljust(10), str(len(b)).rjust(5), ' '.join(f"{k}={v}" for k, v in sorted(c.items())), '  ->', tail[:40])
print('total', dict(tot))
