"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into the per-kernel table rocprofv3 --stats
prints (name, calls, total/avg/min/max us, %), for committing under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3, max(vgpr_count), max(sgpr_count), max(scratch_size), max(lds_size) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    lines = [f"# kernel stats from {db} (durations in microseconds)",
             f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} "
             f"{'vgpr':>5s} {'sgpr':>5s} {'scratch':>8s} {'lds':>6s}"]
    for r in rows:
        lines.append(f"{r[0][:100]:100s} {r[1]:6d} {r[2]:12.1f} {r[3]:10.2f} {r[4]:10.2f} {r[5]:10.2f} {100 * r[2] / tot:6.2f} "
                     f"{r[6]:5d} {r[7]:5d} {r[8]:8d} {r[9]:6d}")
    # per launch size (one row that averages a 65 536-tuple and a 2^20-tuple launch explains neither): kernels that
    # were launched with more than one grid get a row per grid, largest share first
    try:
        cols = [c[1] for c in con.execute("pragma table_info(kernels)").fetchall()]
        gx = next((c for c in cols if c.lower() in ("grid_x", "grid_size_x", "gridx")), None) or next((c for c in cols if "grid" in c.lower() and c.lower().endswith("x")), None)
        if gx:
            per = con.execute(f"select name, {gx}, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, sum(end-start)/1e3 "
                              f"from kernels group by name, {gx} order by 7 desc").fetchall()
            multi = {}
            for r in per:
                multi.setdefault(r[0], []).append(r)
            lines += ["", f"# per launch size ({gx} = work-items of the launch in x): kernels launched with more than one grid",
                      f"{'kernel':70s} {gx:>12s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}"]
            for name, rs in sorted(multi.items(), key=lambda kv: -sum(x[6] for x in kv[1])):
                if len(rs) < 2:
                    continue
                for r in sorted(rs, key=lambda x: -x[1])[:8]:
                    lines.append(f"{name[:70]:70s} {r[1]:12d} {r[2]:6d} {r[3]:10.2f} {r[4]:10.2f} {r[5]:10.2f}")
    except sqlite3.Error as e:  # (an older schema: the aggregate table stands alone)
        lines += ["", f"# per-launch-size table unavailable: {e}"]
    # lone vs overlapped launches (the Merkle pass's table average mixed the two-roots-in-flight launches, which
    # run beside another root's tail, with the lone ones the roofline is quoted on): a launch is "overlapped" when launches of OTHER
    # streams cover more than 10 % of its duration; kernels that have both kinds get a row each
    try:
        evs = con.execute("select name, start, end from kernels order by start").fetchall()
        kinds = {}
        active = []  # (end, index) of launches still running
        cover = [0.0] * len(evs)
        for i, (name, st, en) in enumerate(evs):
            active = [(e, j) for (e, j) in active if e > st]
            for e, j in active:
                ov = min(e, en) - st
                if ov > 0:
                    cover[i] += ov
                    cover[j] += ov
            active.append((en, i))
        for i, (name, st, en) in enumerate(evs):
            dur = max(en - st, 1)
            kinds.setdefault(name, {}).setdefault("overlapped" if cover[i] > 0.1 * dur else "lone", []).append(dur / 1e3)
        both = {k: v for k, v in kinds.items() if len(v) == 2}
        if both:
            lines += ["", "# lone vs overlapped launches (overlapped: other launches run during > 10 % of the launch): kernels that have both",
                      f"{'kernel':70s} {'kind':>11s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s}"]
            for name, v in sorted(both.items(), key=lambda kv: -sum(sum(x) for x in kv[1].values()))[:24]:
                for kind in ("lone", "overlapped"):
                    d = v[kind]
                    lines.append(f"{name[:70]:70s} {kind:>11s} {len(d):6d} {sum(d) / len(d):10.2f} {min(d):10.2f} {max(d):10.2f}")
    except sqlite3.Error as e:
        lines += ["", f"# lone / overlapped table unavailable: {e}"]
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
