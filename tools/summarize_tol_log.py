#!/usr/bin/env python3
"""FGS_TOL_LOG (tests/helpers.py) -> profiles/rNN_gpu_tolerance_slack.txt: per assert site the worst achieved error of `pytest -m gpu`.
Two tables: (1) the max-norm metric of every site (rel_inf / outlier_fraction), (2) the element-wise 1e-4 criterion, three-way: the fraction
of entries beyond 1e-4 |x| + 1e-4 median|x| for the HIP path and for the fp32 oracle, both against the fp64 evaluation of the same formulas,
and directly HIP vs oracle32. usage: python tools/summarize_tol_log.py gpurun_out/tol.log > profiles/archive/r03_gpu_tolerance_slack.txt"""
import collections
import re
import sys

path = sys.argv[1]
maxnorm = collections.defaultdict(lambda: [0, 0.0, 0.0])
elem = collections.defaultdict(lambda: collections.defaultdict(list))
notes = collections.Counter()
masked, ratios = [], {}
for line in open(path):
    m = re.match(r'(\S+) (\S+) (\w+)=(\S+)(.*)', line)
    if not m:
        continue
    site, fn, kind, val, rest = m.groups()
    if kind == 'masked_fraction':
        masked.append((fn, val, rest.strip()))
        continue
    if kind == 'three_way_ratio':
        rows = re.search(r"'rows': (\d+)", rest); fac = re.search(r"'factor': ([\d.]+)", rest); fo = re.search(r"'frac_oracle': '([^']+)'", rest)
        if rows and fac and fo and float(fo.group(1)) > 1e-3:
            key = (fn, 'rows >= 60 k (factor 1.10)' if int(rows.group(1)) >= 60000 else 'smaller scenes (factor 1.25)')
            ratios[key] = max(ratios.get(key, 0.0), float(val))
        continue
    if kind == 'int_mismatch_primitives':
        notes[(site, fn, 'budget branch taken' if int(val) else 'no integer mismatch')] += 1
        continue
    try:
        v = float(val)
    except ValueError:
        continue
    em = re.match(r'elem_(hip_vs_f64|oracle32_vs_f64|hip_vs_oracle32)_(\w+)', kind)
    n = re.search(r' n=(\d+)', rest)
    if em:
        elem[(site, fn, em.group(2), n.group(1) if n else '?')][em.group(1)].append(v)
        continue
    if kind.startswith('elementwise'):
        elem[(site, fn, kind[len('elementwise_'):] or 'tensor', n.group(1) if n else '?')]['direct'].append(v)
        continue
    ri = re.search(r'rel_inf=(\S+)', rest)
    fa = re.search(r'frac_above_1e-4_of_max=(\S+)', rest)
    a = maxnorm[(site, fn)]
    a[0] += 1
    a[1] = max(a[1], float(ri.group(1)) if ri else v)
    a[2] = max(a[2], float(fa.group(1)) if fa else 0.0)

print('# Worst achieved error per assert site of `python -m pytest tests -m gpu` on an MI355X, recorded with FGS_TOL_LOG (tests/helpers.py).')
print('# (1) max-norm: rel_inf = max|a - ref| / max|ref| over the compared tensor (bar: 1e-4)')
print(f'{"site":32s} {"function":52s} {"calls":>6s}  {"worst rel_inf":>13s}  {"frac > 1e-4*max":>15s}')
above = 0
for (site, fn), (calls, worst, frac) in sorted(maxnorm.items()):
    print(f'{site:32s} {fn:52s} {calls:6d}  {worst:13.2e}  {frac:15.2e}')
    above += worst > 1e-4
print(f'# sites above 1e-4: {above} of {len(maxnorm)}')
print()
print('# (2) element-wise: fraction of entries with |a - x| > 1e-4 |x| + 1e-4 median|x| (helpers.elementwise_fraction), outside the oracle\'s')
print('#     threshold-risk masks. x = the fp64 evaluation of the same formulas (oracle.forward_backward_f64). The fp32 oracle itself misses this')
print('#     bar wherever a gradient entry is an ill-conditioned sum; the assert is HIP <= f x oracle32 + 1e-4 (+ 4 sigma of the count), f = 1.10 from 60 k rows on, 1.25 below.')
print(f'{"site":26s} {"tensor":20s} {"entries":>10s}  {"HIP vs fp64":>12s}  {"oracle32 vs fp64":>16s}  {"ratio":>6s}  {"HIP vs oracle32":>15s}')
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for (site, fn, tensor, n), d in sorted(elem.items()):
    if 'hip_vs_f64' in d:
        for i, fh in enumerate(d['hip_vs_f64']):
            fo, fd = d['oracle32_vs_f64'][i], d['hip_vs_oracle32'][i]
            if n != '?' and int(n) >= 2_000_000:
                print(f'{site:26s} {tensor:20s} {n:>10s}  {fh:12.3e}  {fo:16.3e}  {fh / fo if fo else float("nan"):6.2f}  {fd:15.3e}')
            a = agg[(fn, tensor)]
            a[0] += 1; a[1] = max(a[1], fh); a[2] = max(a[2], fo); a[3] = max(a[3], fd)
            a[4] = max(a[4], (fh - 1e-4) / fo if fo > 1e-3 else 0.0)
    else:
        a = agg[(fn, tensor + ' (direct)')]
        a[0] += len(d['direct']); a[3] = max([a[3]] + d['direct'])
print('# all sites, worst per (function, tensor); "worst ratio" over the calls where the oracle misses more than 1e-3 of the entries')
print(f'{"function":48s} {"tensor":28s} {"calls":>6s}  {"HIP vs fp64":>12s}  {"oracle32 vs fp64":>16s}  {"worst ratio":>11s}  {"HIP vs oracle32":>15s}')
for (fn, tensor), a in sorted(agg.items()):
    print(f'{fn:48s} {tensor:28s} {a[0]:6d}  {a[1]:12.3e}  {a[2]:16.3e}  {a[4]:11.2f}  {a[3]:15.3e}')
print()
print('# (3) integer intermediates of _forward_check (screen bounds / tile counts vs the oracle): how often the libm-ULP budget branch is taken')
for (site, fn, what), c in sorted(notes.items()):
    print(f'{site:32s} {fn:40s} {what:24s} {c}')
print()
print('# (4) threshold-risk masks of the flip-aware comparisons: realised masked fraction per call (pixels / Gaussians excluded from the 1e-4 bars), its bound,')
print('#     and the share of Gaussians in the looser "near" class (adversarial fuzz scenes only)')
for fn, val, rest in masked:
    print(f'{fn:48s} masked {val:>10s}   {rest}')
print()
print('# (5) worst HIP / oracle32 miss ratio of the three-way rule per test function (calls where the oracle misses more than 1e-3 of the entries)')
for (fn, cls), r in sorted(ratios.items()):
    print(f'{fn:48s} {cls:32s} {r:6.3f}')
