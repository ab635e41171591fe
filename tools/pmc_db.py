"""Per-kernel averages from rocprofv3 --pmc result databases (development aid).
usage: pmc_db.py 'glob/of/*.db' [kernel-substring]"""
import sqlite3, glob, sys, collections
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_standardRenderer<0,"
for f in sorted(glob.glob(sys.argv[1], recursive=True)):
    c = sqlite3.connect(f)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    rows = c.execute("select * from counters_collection").fetchall()
    agg = collections.defaultdict(list)
    ik = cols.index('kernel_name') if 'kernel_name' in cols else None
    for r in rows:
        d = dict(zip(cols, r))
        if kernel in str(d.get('kernel_name', '')):
            agg[d['counter_name']].append(float(d['value']))
    print("==", f.split('/')[-2])
    print("  " + "  ".join("%s=%.1fM" % (k.replace('SQ_', ''), sum(v) / len(v) / 1e6) for k, v in sorted(agg.items())))
