"""Summarises a rocprofv3 rocpd sqlite database (`--kernel-trace` output) into the per-kernel
table `rocprofv3 --stats` would print: calls, total/avg/min/max duration, share of GPU time.
Usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    rows = db.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    agg = {}
    for name, st, en in rows:
        name = re.sub(r"\s+", " ", name)
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        dur = en - st
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["| kernel | calls | total_us | avg_us | min_us | max_us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append(f"| `{short}` | {a[0]} | {a[1] / 1e3:.1f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | "
                     f"{a[3] / 1e3:.2f} | {100 * a[1] / total:.1f} |")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
