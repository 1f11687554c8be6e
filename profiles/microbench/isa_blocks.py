"""Instruction mix per basic block of one function in a gfx950 .s file (hipcc --save-temps):
    python profiles/microbench/isa_blocks.py file.s <function-name-substring> [min instructions]
Prints, per basic block above the threshold: MFMA / VALU / SALU / LDS / global loads / global stores / scratch / s_waitcnt / s_nop counts."""
import re, sys
path, key = sys.argv[1], sys.argv[2]
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*%s\S*:' % re.escape(key), l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('\t.size') or lines[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end]:
    m = re.match(r'^(\.LBB\S+):', l)
    if m:
        blocks.append((name, cur)); name, cur = m.group(1), []
    elif l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'):
        cur.append(l.strip().split()[0])
blocks.append((name, cur))
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_load') or op.startswith('flat_load') or op.startswith('buffer_load'): return 'gload'
    if op.startswith('global_store') or op.startswith('flat_store') or op.startswith('buffer_store'): return 'gstore'
    if op.startswith('scratch_'): return 'scratch'
    if op == 's_waitcnt': return 'wait'
    if op == 's_nop': return 'nop'
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    return 'other'
tot = {}
for name, ops in blocks:
    c = {}
    for o in ops:
        c[cls(o)] = c.get(cls(o), 0) + 1; tot[cls(o)] = tot.get(cls(o), 0) + 1
    if len(ops) >= thr:
        print('%-14s %5d  ' % (name, len(ops)) + ' '.join('%s %d' % kv for kv in sorted(c.items())))
print('function total', sum(tot.values()), tot)
