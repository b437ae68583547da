#!/usr/bin/env python3
"""Aggregates a rocprofv3 --pmc counter_collection.csv per kernel: sum of each counter over dispatches."""
import csv
import sys
from collections import defaultdict


def main(path):
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            calls[k].add(row["Dispatch_Id"])
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        print("%s  [%d dispatches]" % (k, len(calls[k])))
        for c, v in sorted(acc[k].items()):
            print("    %-28s %.6g" % (c, v))


if __name__ == "__main__":
    main(sys.argv[1])
