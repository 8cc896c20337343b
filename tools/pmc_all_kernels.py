"""Per-kernel table of the counters in a rocprofv3 --pmc database (per launch): python tools/pmc_all_kernels.py <results.db>"""
import collections, os, sqlite3, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles'))
from summarize_rocprof import short
db = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in db.execute('pragma table_info(pmc_events)')]
name_col = 'counter_name' if 'counter_name' in cols else [c for c in cols if 'name' in c][-1]
val_col = 'counter_value' if 'counter_value' in cols else [c for c in cols if 'value' in c][-1]
launches = dict(db.execute('select name, count(*) from kernels group by name').fetchall())
rows = collections.defaultdict(dict)
for k, cn, s in db.execute(f'select name, {name_col}, sum({val_col}) from pmc_events group by name, {name_col}'):
    rows[short(k)][cn] = s / (launches.get(k, 0) or 1)
names = sorted({c for r in rows.values() for c in r})
print(f'{"kernel":56s} ' + ' '.join(f'{n[3:] if n.startswith("SQ_") else n:>15s}' for n in names))
for k, r in sorted(rows.items(), key=lambda kv: -max(kv[1].values())):
    print(f'{k[:56]:56s} ' + ' '.join(f'{r.get(n, 0):15.0f}' for n in names))
