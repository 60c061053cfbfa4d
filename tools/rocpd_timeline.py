"""Dump the kernel timeline of the LAST `n` kernels of a rocprofv3 rocpd (sqlite) trace: start/end relative to the
first one, duration, queue/stream, name.  Used to see which launches of one state root overlap."""
import sqlite3
import sys


def main(db, n=120, out=None):
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    want = [c for c in ("start", "end", "queue_id", "stream_id", "name") if c in cols]
    rows = con.execute(f"select {', '.join(want)} from kernels order by start desc limit {int(n)}").fetchall()[::-1]
    t0 = rows[0][0]
    lines = ["# columns: " + " ".join(want) + f"   (all columns of the view: {' '.join(cols)})"]
    for r in rows:
        d = dict(zip(want, r))
        lines.append(f"{(d['start'] - t0) / 1e3:10.1f} {(d['end'] - t0) / 1e3:10.1f} {(d['end'] - d['start']) / 1e3:9.1f} us  q={d.get('queue_id')} s={d.get('stream_id')}  {d['name'][:90]}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(a[0], int(a[1]) if len(a) > 1 else 120, a[2] if len(a) > 2 else None)
