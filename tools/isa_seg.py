"""Instruction histogram of a line range of one kernel in a hipcc -S listing: python tools/isa_seg.py file.s <name-substring> <first> <last> [dump]"""
import collections, re, sys
s = open(sys.argv[1]).read()
name = [n for n in re.findall(r'^(_Z\w+):', s, flags=re.M) if sys.argv[2] in n][0]
a = s.index(name + ':'); b = s.index('.Lfunc_end', a)
seg = s[a:b].split('\n')[int(sys.argv[3]):int(sys.argv[4]) + 1]
ins = [x.split()[0] for x in seg if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
print(collections.Counter(ins).most_common(70))
if len(sys.argv) > 5:
    print('\n'.join(seg))
