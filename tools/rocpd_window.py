"""Kernel timeline around the LAST launch of a kernel whose name contains `needle` and that ran longer than `min_us`: every kernel
whose execution overlaps [start - before_us, end + after_us] of that launch, times relative to its start.
usage: rocpd_window.py trace.db needle [min_us] [before_us] [after_us]"""
import sqlite3
import sys


def main(db, needle, min_us=1000.0, before_us=2000.0, after_us=15000.0):
    con = sqlite3.connect(db)
    rows = con.execute("select start, end, queue_id, stream_id, name from kernels order by start").fetchall()
    hits = [r for r in rows if needle in r[4] and (r[1] - r[0]) / 1e3 >= min_us]
    if not hits:
        print("no launch of", needle)
        return
    a = hits[-1]
    t0 = a[0]
    lo, hi = a[0] - before_us * 1e3, a[1] + after_us * 1e3
    for s, e, q, st, name in rows:
        if e >= lo and s <= hi:
            print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} us  q={q} s={st}  {name[:70]}")


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], a[1], *[float(x) for x in a[2:5]])
