"""Prints the kernel launch sequence (name, duration us, gap to previous end us) of the last `n` launches of a rocprofv3 rocpd db."""
import sqlite3, sys
sys.path.insert(0, '/root/repo/profiles')
from summarize_rocprof import short
db = sqlite3.connect(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = db.execute('select name, start, end from kernels order by start').fetchall()[-n:]
prev = None
for name, s, e in rows:
    print(f'{short(name)[:70]:70s} {(e - s) / 1e3:8.2f} {((s - prev) / 1e3 if prev else 0):8.2f}')
    prev = e
