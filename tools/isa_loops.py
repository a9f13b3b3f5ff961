"""Instruction histogram of the loops of one kernel in a hipcc -S listing.
usage: python tools/isa_loops.py file.s <mangled-name-substring>"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
key = sys.argv[2]
names = re.findall(r'^(_Z\w+):', s, flags=re.M)
name = [n for n in names if key in n][0]
a = s.index(name + ':')
b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
print(name, len(body), 'lines')
for i, l in enumerate(body):
    m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        st = labels[m.group(1)]
        ins = [x.split()[0] for x in body[st:i + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        c = collections.Counter()
        for x in ins:
            if x.startswith('v_'):
                c['VALU'] += 1
            elif x.startswith('s_'):
                c['SALU'] += 1
            elif x.startswith('ds_'):
                c['DS'] += 1
            elif x.startswith(('global_', 'buffer_', 'flat_')):
                c['VMEM'] += 1
            else:
                c[x] += 1
        print(f"loop {m.group(1)} lines {st}-{i}: {len(ins)} instr {dict(c)}")
        if len(sys.argv) > 3 and sys.argv[3] == 'hist':
            print(collections.Counter(ins).most_common(40))
for l in body[-80:]:
    if 'vgpr_count' in l or 'sgpr_count' in l:
        print(l)
