import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if 'kernel_dispatch' in t and t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {sym} s on d.kernel_id = s.id").fetchall()
h = collections.defaultdict(lambda: collections.Counter())
for n, d in rows:
    n = n.split('(')[0]
    if 'rescnn' not in n: continue
    us = d / 1e3
    b = 5 if us < 5 else 20 if us < 20 else 60 if us < 60 else 120 if us < 120 else 300 if us < 300 else 1000 if us < 1000 else 9999
    h[n][b] += 1
for n in h: print(n, sorted(h[n].items()))
