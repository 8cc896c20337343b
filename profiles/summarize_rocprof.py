"""Turns rocprofv3 (ROCm 7.2, rocpd sqlite output) results into the small text summaries committed under profiles/.

  python profiles/summarize_rocprof.py stats gpurun_out/prof_s2/bench_results.db  > profiles/archive/r01_s2_kernel_stats.txt
  python profiles/summarize_rocprof.py pmc   gpurun_out/pmc_fetch/bench_results.db > profiles/archive/r01_s2_pmc_fetch.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r'rocprim::ROCPRIM_\d+_NS::', 'rocprim::', name)
    m = re.search(r'rocprim::detail::(radix_sort_onesweep_iteration|radix_sort_onesweep_global_offsets|scan_impl|init_lookback_scan_state_kernel)', name)
    if m:
        key = 'u16' if 'unsigned short' in name else 'u32'
        extra = ''
        if 'TouchedFromRec' in name: extra = ' [offsets]'
        if 'BucketsOfRange' in name: extra = ' [buckets]'
        lam = re.findall(r'lambda\(auto:1\)#(\d)', name)
        return f'rocprim::{m.group(1)}<{key}>{extra}' + (f' #{lam[-1]}' if 'global_offsets' in name and lam else '')
    name = re.sub(r'\(.*', '', name)
    return name[:90]


def stats(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
                      "max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    # Since round 4 one bench.py run also TRAINS a small model (`trained_like`: thousands of launches at 0.1-0.2 M Gaussians), so a kernel's plain average
    # mixes sizes. The last two columns restrict it to the launches with the kernel's LARGEST grid -- the headline workload's (and the layered scene's,
    # which has the same Gaussian count) -- which is what the bench line's `roofline.avg_kernel_ms` is to be compared with.
    at_max = {}
    for n, gx, c, a in db.execute("select name, grid_x, count(*), avg(duration) from kernels group by name, grid_x"):
        if n not in at_max or gx > at_max[n][0]:
            at_max[n] = (gx, c, a)
    total = sum(r[2] for r in rows)
    print(f'{"kernel":72s} {"calls":>6s} {"total_us":>10s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} {"%":>6s} {"vgpr":>5s} {"agpr":>5s} {"sgpr":>5s} {"lds":>6s} {"scr":>4s} {"grid":>9s} {"wg":>4s} {"calls@max":>9s} {"avg@max_us":>10s}')
    for n, c, s, a, mn, mx, vg, ag, sg, lds, scr, gx, wx in rows:
        print(f'{short(n):72s} {c:6d} {s / 1e3:10.1f} {a / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / total:6.2f} {vg or 0:5d} {ag or 0:5d} {sg or 0:5d} {lds or 0:6d} {scr or 0:4d} {gx or 0:9d} {wx or 0:4d} {at_max[n][1]:9d} {at_max[n][2] / 1e3:10.2f}')


def pmc(path):
    db = sqlite3.connect(path)
    cols = [d[1] for d in db.execute('pragma table_info(pmc_events)')]
    name_col = 'counter_name' if 'counter_name' in cols else [c for c in cols if 'name' in c][-1]
    val_col = 'counter_value' if 'counter_value' in cols else [c for c in cols if 'value' in c][-1]
    kcol = 'name'
    # SQ_* counters come as one row per shader engine and dispatch, TCC-derived ones (FETCH_SIZE, WRITE_SIZE: KiB) as one row per
    # dispatch: total per launch = sum over the rows / number of dispatches of that kernel in the kernel trace
    launches = dict(db.execute('select name, count(*) from kernels group by name').fetchall())
    q = f"select {kcol}, {name_col}, count(*), sum({val_col}) from pmc_events group by {kcol}, {name_col} order by sum({val_col}) desc"
    print(f'{"kernel":72s} {"counter":>14s} {"dispatches":>10s} {"total/dispatch":>18s}')
    for k, cn, c, s in db.execute(q):
        n = launches.get(k, 0) or 1
        print(f'{short(k):72s} {cn:>14s} {n:10d} {s / n:18.1f}')


if __name__ == '__main__':
    {'stats': stats, 'pmc': pmc}[sys.argv[1]](sys.argv[2])
