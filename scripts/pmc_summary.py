import sqlite3, sys, glob, collections
out = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "solve_kernel"
tot = collections.OrderedDict()
for db in sorted(glob.glob(out + "/p*/p_results.db")):
    d = sqlite3.connect(db)
    for name, val, n in d.execute("select counter_name, sum(value), count(*) from counters_collection where kernel_name like ? group by counter_name", (f"%{kern}%",)):
        tot[name] = (val / n)
for k, v in tot.items():
    print(f"{k:28s} {v:18.4e}")
