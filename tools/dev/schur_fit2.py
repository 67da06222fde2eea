# second fit of the Schur tile kernel's per-workgroup stamps (-DSCHUR_STAMPS build whose schedule carries pair:groups:cost):
# duration ~ c0 + c_seg * segments + c_grp * groups + c_f * (fragment products, from the model cost) [+ second slot], any shape
# usage: python tools/dev/schur_fit2.py log.txt [load term of the cost model, default 36]
import re, sys, numpy as np
txt = open(sys.argv[1]).read()
L = int(sys.argv[2]) if len(sys.argv) > 2 else 36
for blk in txt.split('SCHUR workgroups')[1:][-2:]:
    w = {int(a): (float(b), float(c), int(d)) for a, b, c, d, h in re.findall(r'\[(\d+) ([\d.]+) ([\d.]+) (\d+) ([0-9a-f]+)\]', blk.split('SCHUR schedule')[0])}
    sch = {}
    for m in re.finditer(r'\{(\d+)((?: \d+:\d+:\d+)*)\}', blk.split('SCHUR schedule')[1].split('SCHUR stamps')[0]):
        sch[int(m.group(1))] = [tuple(int(v) for v in x.split(':')) for x in m.group(2).split()]
    rows, y, pred_model = [], [], []
    for i, s in sch.items():
        if i not in w or not s: continue
        groups = sum(g for _, g, _ in s); cost = sum(c for _, _, c in s)
        f10 = cost - L * 4 * groups   # ~ 10 x (fragment products per entry), entries ~ 4 x groups
        rows.append([1.0, len(s), groups, f10 / 40.0, 1.0 if i >= 256 else 0.0]); y.append(w[i][1] - w[i][0]); pred_model.append(cost + 4900 * len(s))
    A = np.array(rows); y = np.array(y)
    x, *_ = np.linalg.lstsq(A, y, rcond=None)
    print('n %d  coef us: const %.2f seg %.2f group %.4f product/group %.4f second %.2f | rms %.2f us; dur mean %.1f max %.1f min %.1f' % (len(y), *x, np.sqrt(np.mean((A @ x - y) ** 2)), y.mean(), y.max(), y.min()))
    print('   load term that fits: %.1f (model %d); segment in units: %.0f' % (10 * x[2] / max(x[3], 1e-9) / 4 * 4 / 4, L, x[1] / max(x[3], 1e-9) * 10 / 4 * 4))
    ex = np.array([w[i][1] for i in sch if i in w and sch[i]])
    print('   exit mean %.1f max %.1f; model cost vs duration corr %.3f' % (ex.mean(), ex.max(), np.corrcoef(np.array(pred_model), y)[0, 1]))
