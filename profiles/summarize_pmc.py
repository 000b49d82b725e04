#!/usr/bin/env python3
"""Per-kernel, per-dispatch averages of rocprofv3 --pmc passes (counter_collection CSVs), one line per kernel and pass.

usage: summarize_pmc.py DIR [DIR ...]   (each DIR = the -d output directory of one rocprofv3 --pmc run)
"""
import csv
import sys
from collections import defaultdict
from pathlib import Path


def main(dirs):
    for d in dirs:
        files = sorted(Path(d).rglob("*counter_collection.csv"))
        if not files:
            print(f"{d}: no counter_collection.csv")
            continue
        # kernel -> counter -> dispatch id -> value (a counter row per dispatch and counter; sum over dimensions)
        acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
        for f in files:
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name") or row.get("Kernel Name") or "?"
                    disp = row.get("Dispatch_Id") or row.get("Dispatch Id") or "0"
                    acc[name][row["Counter_Name"]][disp] += float(row["Counter_Value"])
        for name in sorted(acc, key=lambda n: -sum(sum(v.values()) for v in acc[n].values())):
            if not name.startswith("void lchd") and not name.startswith("lchd") and "k_floor" not in name:  # (k_floor: profiles/ubench/dense_floor.hip)
                continue
            counters = acc[name]
            n_disp = max(len(v) for v in counters.values())
            per = {c: round(sum(v.values()) / max(len(v), 1)) for c, v in sorted(counters.items())}
            short = name.rsplit("(", 1)[0] if name.endswith(")") else name  # (drop the argument list, keep "(anonymous namespace)::")
            print(f"{short} | dispatches {n_disp} | per-dispatch: {per}")


if __name__ == "__main__":
    main(sys.argv[1:])
