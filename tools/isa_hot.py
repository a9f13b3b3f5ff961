"""Hot plane loop of a render kernel in a hipcc -S listing: the innermost loop that holds the LDS-DMA (or the taps).
usage: python tools/isa_hot.py file.s <mangled-name-substring> [print]
Prints the instruction classes of the loop (fast fp32 / slow VALU / moves / SALU / DS / VMEM / scratch) and, with `print`, the listing."""
import collections, re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
names = re.findall(r'^(_Z\w+):', s, flags=re.M)
name = [n for n in names if key in n][0]
a = s.index(name + ':'); b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\S*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
hot = [lp for lp in loops if any('s_barrier' in x for x in body[lp[0]:lp[1]]) and any(('lds' in x and 'buffer_load' in x) or 'ds_read_u16_d16_hi' in x or 'ds_read2_b32' in x or 'ds_read2_b64' in x for x in body[lp[0]:lp[1]])]
hot.sort(key=lambda lp: lp[1] - lp[0])
st, en = hot[0]
FAST = ('v_fma_f32', 'v_fmac_f32', 'v_mul_f32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32', 'v_and_b32', 'v_or_b32', 'v_fmaak_f32', 'v_fmamk_f32', 'v_xor_b32')
c = collections.Counter(); hist = collections.Counter()
for x in body[st:en + 1]:
    if not x.startswith('\t') or x.strip().startswith(('.', ';')): continue
    op = x.split()[0]
    hist[op] += 1
    base = re.sub(r'_e32$|_e64$|_sdwa$|_dpp$', '', op)
    if 'scratch_' in op: c['scratch'] += 1
    elif base == 'v_mov_b32': c['v_mov'] += 1
    elif base in FAST: c['valu_fast'] += 1
    elif op.startswith('v_'): c['valu_slow'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['ds'] += 1
    elif op.startswith(('buffer_', 'global_', 'flat_')): c['vmem'] += 1
print(name)
print(f"hot loop lines {st}-{en}: {dict(c)}  (all paths of the loop body, cold ones included)")
print(hist.most_common(60))
if len(sys.argv) > 3:
    for i in range(st, en + 1):
        l = body[i]
        if l.strip().startswith(';') and 'ASM' not in l: continue
        print(i, l)
