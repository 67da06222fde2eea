# fits the Schur tile kernel's per-workgroup entry / exit stamps (a -DSCHUR_STAMPS build, tools/dev/r05_schur_stamps.sh) to
# segments + groups by pair type, per workgroup and per CU: the cost model of the split in ba_prepare_impl (bundle.hip)
# usage: python tools/dev/schur_fit.py gpurun_out/r05_schur_stamps/log.txt [free cameras]
import re, sys, numpy as np
txt = open(sys.argv[1]).read()
blocks = txt.split('SCHUR workgroups')[1:]
F = int(sys.argv[2]) if len(sys.argv) > 2 else 49
TC = 8
nt = (F + TC - 1)//TC
def frags(t):
    n = min(TC, F - t*TC); return 1 if n <= 2 else (2 if n <= 5 else 3)
def ptype(pr):
    a = 0
    while (a+1)*(a+2)//2 <= pr: a += 1
    b = pr - a*(a+1)//2
    return ('D' if a == b else 'O') + str(frags(a)) + str(frags(b))
res = []
for blk in blocks[-3:]:
    w = {int(a): (float(b), float(c), int(d)) for a, b, c, d, h in re.findall(r'\[(\d+) ([\d.]+) ([\d.]+) (\d+) ([0-9a-f]+)\]', blk.split('SCHUR schedule')[0])}
    sch = {}
    for m in re.finditer(r'\{(\d+)((?: \d+:\d+)*)\}', blk.split('SCHUR schedule')[1].split('SCHUR stamps')[0]):
        sch[int(m.group(1))] = [(int(x.split(':')[0]), int(x.split(':')[1])) for x in m.group(2).split()]
    types = sorted({ptype(p) for s in sch.values() for p, g in s})
    rows, y = [], []
    for i, s in sch.items():
        if i not in w or not s: continue
        r = [len(s)] + [sum(g for p, g in s if ptype(p) == t) for t in types] + [1.0 if i >= 256 else 0.0]
        rows.append(r); y.append(w[i][1] - w[i][0])
    A = np.array(rows, float); y = np.array(y)
    x, *_ = np.linalg.lstsq(A, y, rcond=None)
    pred = A @ x
    print('types', ['seg'] + types + ['second'], 'n', len(y))
    print(' coef us', np.round(x, 4), ' resid rms %.2f us; y mean %.1f max %.1f min %.1f' % (np.sqrt(np.mean((pred-y)**2)), y.mean(), y.max(), y.min()))
    print(' per group relative to O33:', {t: round(x[1+k]/x[1+types.index('O33')], 3) for k, t in enumerate(types)}, 'seg in O33 groups: %.1f' % (x[0]/x[1+types.index('O33')]))
print('---- per CU (blocks b and b+256 share a CU)')
for blk in blocks[-3:]:
    w = {int(a): (float(b), float(c), int(d)) for a, b, c, d, h in re.findall(r'\[(\d+) ([\d.]+) ([\d.]+) (\d+) ([0-9a-f]+)\]', blk.split('SCHUR schedule')[0])}
    sch = {}
    for m in re.finditer(r'\{(\d+)((?: \d+:\d+)*)\}', blk.split('SCHUR schedule')[1].split('SCHUR stamps')[0]):
        sch[int(m.group(1))] = [(int(x.split(':')[0]), int(x.split(':')[1])) for x in m.group(2).split()]
    types = sorted({ptype(p) for s in sch.values() for p, g in s})
    rows, y = [], []
    for i in range(256):
        s = sch.get(i, []) + sch.get(i + 256, [])
        if not s: continue
        r = [1.0, len(s)] + [sum(g for p, g in s if ptype(p) == t) for t in types]
        rows.append(r); y.append(max(w[j][1] for j in (i, i + 256) if j in w))
    A = np.array(rows, float); y = np.array(y)
    x, *_ = np.linalg.lstsq(A, y, rcond=None)
    pred = A @ x
    print('types', ['const', 'seg'] + types, 'n', len(y))
    print(' coef us', np.round(x, 4), ' resid rms %.2f us; y mean %.1f max %.1f min %.1f' % (np.sqrt(np.mean((pred-y)**2)), y.mean(), y.max(), y.min()))
    o = x[2+types.index('O33')]
    print(' per group relative to O33:', {t: round(x[2+k]/o, 3) for k, t in enumerate(types)}, 'seg in O33 groups: %.1f' % (x[1]/o), 'const %.1f' % (x[0]/o))
print('---- last block: CUs sorted by finish')
cu = []
for i in range(256):
    s0, s1 = sch.get(i, []), sch.get(i + 256, [])
    if i not in w: continue
    cu.append((max(w[j][1] for j in (i, i + 256) if j in w), i, i % 8, i // 8, [(p, ptype(p), g) for p, g in s0], [(p, ptype(p), g) for p, g in s1], w[i][1], w.get(i+256, (0,0,0))[1]))
cu.sort()
for c in cu[:12] + cu[-24:]: print(c)
import collections
print('by XCD mean finish', [round(np.mean([c[0] for c in cu if c[2] == x]), 1) for x in range(8)])
print('by XCD max finish', [round(np.max([c[0] for c in cu if c[2] == x]), 1) for x in range(8)])
