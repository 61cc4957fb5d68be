"""Attribute the instructions of a kernel to source lines: parse `hipcc -S -gline-tables-only` output (.loc
directives) and count VALU / SALU / LDS / VMEM instructions per (file, line).  Static counts — loops count once.
usage: isa_by_line.py file.s <kernel-substring> [file-substring [first_line last_line]]
       isa_by_line.py --meta file.s      registers and spills of every kernel of the file (the code object's metadata:
                                         .vgpr_count, .vgpr_spill_count, .sgpr_count, .sgpr_spill_count, scratch bytes, kernarg bytes)
Make file.s for ONE instantiation of the obs kernel with
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I marlgrid_amd/csrc -DMG_DEV_ONLY="7,8,16,0,0" -S -gline-tables-only \
        --cuda-device-only marlgrid_amd/csrc/mg_render.hip -o file.s"""
import re, sys, collections
if sys.argv[1] == "--meta":
    name, row = None, {}
    for line in open(sys.argv[2]):
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m and ".kd" not in m.group(1):
            name = m.group(1)
        m = re.match(r"\s*\.(vgpr_count|vgpr_spill_count|sgpr_count|sgpr_spill_count|private_segment_fixed_size|kernarg_segment_size):\s+(\d+)", line)
        if m:
            row[m.group(1)] = int(m.group(2))
        if line.strip().startswith(".wavefront_size") and row:
            print("%s\n   vgpr %d (spilled %d)  sgpr %d (spilled %d)  scratch %d B  kernarg %d B" % (
                name, row.get("vgpr_count", -1), row.get("vgpr_spill_count", -1), row.get("sgpr_count", -1),
                row.get("sgpr_spill_count", -1), row.get("private_segment_fixed_size", -1), row.get("kernarg_segment_size", -1)))
            row = {}
    sys.exit(0)
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
