"""Prints the kernel timeline of one discriminator update and one PPO minibatch from a rocprofv3
rocpd database: per kernel the start offset, duration and the idle gap since the previous kernel
on the same queue. Usage: python tools/rocpd_timeline.py <results.db> [anchor-substring] [occurrence]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\s+", " ", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def main(path, anchor="gather_concat", occ=20, count=40):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"s.{name_col}, d.start, d.end" + (f", d.{qcol}" if qcol else ", 0")
    rows = db.execute(f"select {sel} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    hits = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(hits) <= occ:
        print("anchor not found often enough:", len(hits))
        return
    i0 = hits[occ]
    q = rows[i0][3]
    seq = [r for r in rows[i0:] if r[3] == q][:count]
    t0 = seq[0][1]
    prev_end = None
    print(f"queue {q}; anchor '{anchor}' occurrence {occ}")
    print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7}  kernel")
    for name, st, en, _ in seq:
        gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.2f} {gap:7.2f}  {short(name)}")
        prev_end = en


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "gather_concat", int(a[3]) if len(a) > 3 else 20, int(a[4]) if len(a) > 4 else 40)
