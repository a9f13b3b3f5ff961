"""Sanity check for the hand-scheduled box loads of render_lds.hip (inline-asm buffer loads the compiler does not
track): in a `hipcc -S` listing, report every instruction that touches the destination registers of such a load
between the load and the next explicit `s_waitcnt vmcnt` inside an asm block.  Linear scan (ignores control flow):
a hit must be inspected by hand.   usage: python tools/check_async_loads.py file.s <mangled-name-substring>"""
import re
import sys

s = open(sys.argv[1]).read()
names = re.findall(r'^(_Z\w+):', s, flags=re.M)
bad = 0
for name in [n for n in names if sys.argv[2] in n]:
    a = s.index(name + ':')
    body = s[a:s.index('.Lfunc_end', a)].split('\n')
    in_app = False
    loads = []  # (line, regs)
    for i, l in enumerate(body):
        t = l.strip()
        if t.startswith(';;#ASMSTART'):
            in_app = True
        elif t.startswith(';;#ASMEND'):
            in_app = False
        m = re.match(r'buffer_load_dwordx4 v\[(\d+):(\d+)\]', t)
        if m and in_app:
            loads.append((i, set(range(int(m.group(1)), int(m.group(2)) + 1))))
    for i, regs in loads:
        for j in range(i + 1, len(body)):
            t = body[j].strip()
            if not t or t.startswith((';', '.')) or t.startswith('s_waitcnt vmcnt') and not body[j - 1].strip().startswith(';;#ASMSTART'):
                continue
            if t.startswith('s_waitcnt vmcnt') and body[j - 1].strip().startswith(';;#ASMSTART'):
                break
            used = set()
            for m in re.finditer(r'v\[(\d+):(\d+)\]', t):
                used |= set(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r'\bv(\d+)\b', t):
                used.add(int(m.group(1)))
            if used & regs and not (t.startswith('buffer_load_dwordx4') and j in [x for x, _ in loads]):
                print(f'{name}: load@{i} regs v{min(regs)}-{max(regs)} touched @{j}: {t[:80]}')
                bad += 1
                break
print('hits:', bad)
