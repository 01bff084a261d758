#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel and prints a markdown table.
usage: pmc_summary.py <label>=<csv> ... [--match SUBSTR] [--per N]   (--per: divide by N, e.g. wave-steps)"""
import collections
import csv
import sys


def main():
    args, match, per = [], "", 1.0
    it = iter(sys.argv[1:])
    for a in it:
        if a == "--match":
            match = next(it)
        elif a == "--per":
            per = float(next(it))
        else:
            args.append(a)
    rows, names = [], []
    for a in args:
        label, path = a.split("=", 1)
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(path)):
            if match in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(r["Kernel_Name"], r["Counter_Name"])] += 1
        for k, d in acc.items():
            rows.append((label, k, d, max(n[(k, c)] for c in d)))
            names += [c for c in d if c not in names]
    print("| run | kernel | launches | " + " | ".join(names) + " |")
    print("|---|---|---|" + "---|" * len(names))
    for label, k, d, launches in rows:
        print(f"| {label} | `{k[:60]}` | {launches} | " + " | ".join(f"{d.get(c, 0) / per:.4g}" for c in names) + " |")


if __name__ == "__main__":
    main()
