#!/usr/bin/env python3
"""summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean duration and counters"""
import collections
import csv
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("ansel::", "")
    m = re.match(r"(?:void\s+)?([A-Za-z_0-9]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def main(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    names = []
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if r["Counter_Name"] not in names:
            names.append(r["Counter_Name"])
    print("%-34s %8s " % ("kernel", "ms") + " ".join("%14s" % n.replace("SQ_", "") for n in names))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(dur[kv[0]].values())):
        d = dur[k]
        print("%-34s %8.3f " % (k, sum(d.values()) / len(d))
              + " ".join("%14.4g" % (sum(v[n]) / len(v[n])) if n in v else "%14s" % "-" for n in names))


if __name__ == "__main__":
    main(sys.argv[1])
