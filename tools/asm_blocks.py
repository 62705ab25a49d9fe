"""Instruction counts per basic block of one kernel in a hipcc -S dump: usage asm_blocks.py file.s mangled-name-prefix [min]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(txt) if l.startswith(sys.argv[2])][0]
end = [i for i in range(start, len(txt)) if 's_endpgm' in txt[i]][0]
lines = txt[start:end]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 12
blocks = []; cur = None
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        cur = [m.group(1), i, 0, 0, 0, 0]; blocks.append(cur); continue
    if cur is None: continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if op.startswith('v_'): cur[2] += 1
    elif op.startswith('s_'): cur[3] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cur[4] += 1
    elif op.startswith('ds_'): cur[5] += 1
print("block, line, VALU, SALU, VMEM, LDS")
for b in blocks:
    if sum(b[2:]) >= mn: print(b)
tot = [sum(b[k] for b in blocks) for k in range(2, 6)]
print("total", tot)
for l in lines:
    if 'NumVgprs' in l or 'ScratchSize' in l: print(l)
for l in txt[end:end + 80]:
    if 'NumVgprs:' in l or 'ScratchSize:' in l: print(l.strip())
