"""Per-kernel average of a PMC counter from a rocprofv3 rocpd database. Usage: rocpd_pmc.py db"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
t = lambda p: next(x for x in tabs if x.startswith(p))
kd, ks, pe, pi = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
pcols = [r[1] for r in db.execute(f"pragma table_info({pe})")]
icols = [r[1] for r in db.execute(f"pragma table_info({pi})")]
scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name_col = "kernel_name" if "kernel_name" in scols else "display_name"
q = (f"select s.{name_col}, i.name, e.value from {pe} e join {pi} i on e.pmc_id = i.id "
     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
agg = {}
for kname, cname, val in db.execute(q):
    a = agg.setdefault((re.sub(r"\s+", " ", kname)[:90], cname), [0, 0.0])
    a[0] += 1; a[1] += val
for (k, c), (n, v) in sorted(agg.items()):
    print(f"{k:92s} {c:12s} n={n:4d} avg={v / n:14.1f}")
