"""Per-kernel mean of the PMC counters in a rocprofv3 --pmc run (csv output: *_counter_collection.csv)."""
import csv
import glob
import sys
from collections import defaultdict


def main(root, out=None):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                acc[row["Kernel_Name"][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = ["# per-kernel mean of the PMC counters per dispatch (rocprofv3 --pmc); FETCH_SIZE / WRITE_SIZE are in KiB as reported"]
    for k in sorted(acc, key=lambda k: -sum(sum(v) for v in acc[k].values())):
        for c, v in sorted(acc[k].items()):
            lines.append(f"{k:80s} {c:12s} dispatches {len(v):5d}  mean {sum(v) / len(v):16.1f}  max {max(v):16.1f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
