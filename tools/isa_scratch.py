"""Where a kernel's scratch (spill) accesses sit: python tools/isa_scratch.py file.s <name-substring>
Lists every loop (back edge) with its barrier / tap / DMA / scratch counts; the plane loop is the innermost one holding an s_barrier."""
import re, sys
s = open(sys.argv[1]).read()
names = re.findall(r'^(_Z\w+):', s, flags=re.M)
name = [n for n in names if sys.argv[2] in n][0]
a = s.index(name + ':'); b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
print(name, 'total scratch ops', sum('scratch_' in x for x in body))
for i, l in enumerate(body):
    m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i]
        nb = sum('s_barrier' in x for x in seg)
        if nb: print(labels[m.group(1)], i, 'barriers', nb, 'taps', sum('ds_read_u16_d16_hi' in x or 'ds_read2_b32' in x for x in seg), 'dma', sum('buffer_load_dwordx4' in x for x in seg), 'scratch', sum('scratch_' in x for x in seg), 'valu', sum(x.strip().startswith('v_') for x in seg))
