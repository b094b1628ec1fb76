import re, sys
from collections import Counter
path, kname = sys.argv[1], sys.argv[2]
s = open(path).read()
m = re.search(r'(\n' + kname + r'[^\n]*:[^\n]*\n)(.*?)\.Lfunc_end\d+', s, re.S)
body = m.group(2)
blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
best = None
for i in range(1, len(blocks), 2):
    lab, code = blocks[i], blocks[i + 1]
    n = len([l for l in code.split('\n') if l.strip() and not l.strip().startswith(';')])
    if best is None or n > best[1]:
        best = (lab, n, code)
lab, n, code = best
print("largest block", lab, n, "instructions")
c = Counter()
for l in code.split('\n'):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    c[l.split()[0]] += 1
for k, v in c.most_common():
    print("%5d %s" % (v, k))
md = re.search(r'\.name:\s+' + kname + r'.*?\n(.*?)(?=\n  - \.|\Z)', s, re.S)
for key in ('vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size'):
    mm = re.search(r'\.%s:\s*(\d+)' % key, s[s.find('.name:           ' + kname):][:3000])
    if mm: print(key, mm.group(1))
