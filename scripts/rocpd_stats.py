"""Summarise a rocprofv3 rocpd sqlite database (kernel trace): per-kernel count/total/avg, like
`rocprofv3 --stats` CSV output.  usage: python scripts/rocpd_stats.py results.db [--by-grid] [--top N]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[1])
    gx = "grid_size_x" if "grid_size_x" in cols else None
    q = f"select s.{name_col}, d.start, d.end" + (f", d.{gx}, d.grid_size_y, d.grid_size_z" if gx else "") + \
        f" from {kd} d join {ks} s on d.kernel_id = s.id"
    agg = {}
    total = 0
    for row in cur.execute(q):
        name, s, e = row[0], row[1], row[2]
        key = name.split("(")[0]
        if len(key) > 110:
            key = key[:110]
        if by_grid and gx:
            key += f"  grid=({row[3]},{row[4]},{row[5]})"
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += e - s
        total += e - s
    print(f"{'kernel':<130} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:<130} {n:>7} {t / 1e6:>10.3f} {t / n / 1e3:>10.2f} {100.0 * t / total:>6.2f}")
    print(f"TOTAL kernel time {total / 1e6:.3f} ms")


if __name__ == "__main__":
    main()
