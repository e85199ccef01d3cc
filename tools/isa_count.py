"""Instruction histogram of one kernel in a hipcc -S listing: python tools/isa_count.py file.s <substring of mangled name>"""
import collections, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = [l.split(':')[0] for l in s.split('\n') if key in l and l.startswith('_Z') and ':' in l]
for name in names[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    i = s.index('\n' + name + ':')
    j = s.index('.Lfunc_end', i)
    ops = collections.Counter()
    for line in s[i:j].split('\n'):
        line = line.strip()
        if not line or line.startswith(('.', ';', '/')) or line.endswith(':'):
            continue
        ops[line.split()[0]] += 1
    tot = sum(ops.values())
    cls = collections.Counter()
    for k, v in ops.items():
        c = ('trans' if k.startswith(('v_exp', 'v_rcp', 'v_log', 'v_rsq', 'v_sqrt')) else 'pk' if k.startswith('v_pk') else
             'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else 'vmem' if k.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else
             'lds' if k.startswith('ds_') else 'other')
        cls[c] += v
    print(name, 'total', tot, dict(cls))
    print('  ' + ', '.join(f'{k}:{v}' for k, v in ops.most_common(28)))
