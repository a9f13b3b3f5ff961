"""VGPRs that live through a line range of a hipcc -S listing (read in the range, never fully defined before their first read there).
usage: python tools/isa_livein.py file.s <name-substring> <first-line> <last-line>"""
import re, sys
s = open(sys.argv[1]).read()
names = re.findall(r'^(_Z\w+):', s, flags=re.M)
name = [n for n in names if sys.argv[2] in n][0]
a = s.index(name + ':'); b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
lo, hi = int(sys.argv[3]), int(sys.argv[4])
def regs(tok):
    out = []
    for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
defined, livein, reads = set(), {}, {}
for i in range(lo, hi + 1):
    l = body[i]
    if not l.startswith('\t') or l.strip().startswith(('.', ';')): continue
    parts = l.strip().split(None, 1)
    if len(parts) < 2: continue
    op, args = parts
    args = args.split(';')[0]
    ops = [x.strip() for x in args.split(',')]
    if op.startswith(('v_', 'ds_read', 'buffer_load', 'scratch_load', 'global_load')) and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')) and 'lds' not in args:
        dst, src = ops[:1], ops[1:]
    else:
        dst, src = [], ops
    for t in src:
        for r in regs(t):
            if r not in defined and r not in livein: livein[r] = i
            reads[r] = reads.get(r, 0) + 1
    for t in dst:
        for r in regs(t): defined.add(r)
print(len(livein), 'live-in VGPRs:', sorted(livein.items()))
