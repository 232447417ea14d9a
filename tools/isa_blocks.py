"""Developer tool: per-basic-block instruction statistics of one kernel in a hipcc -S dump."""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
s = open(path).read()
i = s.index(name + ':')
body = s[i:s.index('s_endpgm', i)]
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write(body)
lines = body.split('\n')
blocks = []
cur = None
for idx, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), idx, collections.Counter()]
        blocks.append(cur)
    elif cur is not None:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'):
            cur[2][t.split()[0]] += 1
tot = collections.Counter()
for b in blocks:
    c = b[2]
    tot.update(c)
    nm = sum(v for k, v in c.items() if k.startswith('v_mfma'))
    if nm > 0:
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        acc = sum(v for k, v in c.items() if k.startswith('v_accvgpr'))
        ds = sum(v for k, v in c.items() if k.startswith('ds_'))
        salu = sum(v for k, v in c.items() if k.startswith('s_'))
        print(f"{b[0]:12s} line {b[1]:5d} n {sum(c.values()):4d} mfma {nm:3d} valu {valu:4d} (acc {acc:3d} exp {c['v_exp_f32_e32']:3d} "
              f"pk {sum(v for k,v in c.items() if k.startswith('v_pk_')):3d}) ds {ds:3d} salu {salu:3d} nop {c['s_nop']:3d} vm {c['global_load_lds_dwordx4']}")
print("total", sum(tot.values()), tot.most_common(12))
