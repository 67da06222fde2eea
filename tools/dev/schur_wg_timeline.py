# reads the "SCHUR workgroups" block printed by a -DSCHUR_STAMPS build (tools/dev/r05_schur_stamps.sh): exits by segment count and by CU
import re, sys, statistics as S, collections
t = open(sys.argv[1]).read().split('SCHUR stamps')[0]
w = [(int(a), float(b), float(c), int(d), int(h, 16)) for a, b, c, d, h in re.findall(r'\[(\d+) ([\d.]+) ([\d.]+) (\d+) ([0-9a-f]+)\]', t)]
print(len(w), t.splitlines()[0][:60])
for s in (1, 2, 3, 4):
    q = [x for x in w if x[3] == s]
    if q: print(' segments', s, 'n', len(q), 'exit med %.1f min %.1f max %.1f' % (S.median([x[2] for x in q]), min(x[2] for x in q), max(x[2] for x in q)))
cu = collections.defaultdict(list)
for x in w:
    hw = x[4]; cu[(hw >> 16, (hw >> 13) & 7, (hw >> 8) & 15)].append(x)   # (xcc, se, cu)
print(' distinct CUs', len(cu), 'workgroups per CU', collections.Counter(len(v) for v in cu.values()))
fin = sorted(max(x[2] for x in v) for v in cu.values())
print(' CU finish: min %.1f med %.1f max %.1f' % (fin[0], S.median(fin), fin[-1]))
if len(sys.argv) > 2:
    for k, v in sorted(cu.items(), key=lambda kv: max(x[2] for x in kv[1]))[-8:]: print('  late', k, [(x[0], x[0] // 8, x[2], x[3]) for x in v])
    for k, v in sorted(cu.items(), key=lambda kv: max(x[2] for x in kv[1]))[:8]: print('  early', k, [(x[0], x[0] // 8, x[2], x[3]) for x in v])
