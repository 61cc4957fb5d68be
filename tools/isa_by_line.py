"""Attribute the instructions of a kernel to source lines: parse `hipcc -S -gline-tables-only` output (.loc
directives) and count VALU / SALU / LDS / VMEM instructions per (file, line).  Static counts — loops count once.
usage: isa_by_line.py file.s <kernel-substring> [file-substring [first_line last_line]]"""
import re, sys, collections
path, kern = sys.argv[1], sys.argv[2]
fsel = sys.argv[3] if len(sys.argv) > 3 else None
lo = int(sys.argv[4]) if len(sys.argv) > 4 else 0
hi = int(sys.argv[5]) if len(sys.argv) > 5 else 1 << 30
files = {}
cur = None
inside = False
counts = collections.defaultdict(lambda: collections.Counter())
for line in open(path):
    s = line.strip()
    m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
        continue
    if re.match(r'^[A-Za-z_.$][\w.$]*:', s) and not s.startswith('.L'):
        inside = kern in s
        continue
    if not inside:
        continue
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', s)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r'^(v_|s_|ds_|global_|flat_|buffer_|scratch_)(\w+)', s)
    if not m or cur is None:
        continue
    kind = {'v_': 'valu', 's_': 'salu', 'ds_': 'lds', 'global_': 'vmem', 'flat_': 'vmem', 'buffer_': 'vmem', 'scratch_': 'scr'}[m.group(1)]
    if s.startswith('s_waitcnt'): kind = 'wait'
    if s.startswith('s_cbranch') or s.startswith('s_branch'): kind = 'br'
    counts[cur][kind] += 1
tot = collections.Counter()
rows = []
for (f, l), c in counts.items():
    if fsel and fsel not in f: continue
    if not (lo <= l <= hi): continue
    rows.append((f, l, c)); tot.update(c)
rows.sort()
for f, l, c in rows:
    print("%-18s %5d  %s" % (f, l, ' '.join('%s=%d' % kv for kv in sorted(c.items()))))
print("TOTAL", dict(tot), "sum", sum(tot.values()))
