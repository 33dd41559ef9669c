"""Aggregate rocprofv3 --pmc output (one *_counter_collection.csv per pass) into per-kernel SUM and per-launch MEAN.

usage: python3 profiles/aggregate_pmc.py PASSNAME:DIR [PASSNAME:DIR ...] > profiles/rNN_pmc_....csv
"""
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name).strip()
    return name


def main():
    print("pass,kernel,counter,launches,sum,mean_per_launch")   # FETCH_SIZE / WRITE_SIZE are in KB, SQ_* counters in their own units
    for arg in sys.argv[1:]:
        tag, d = arg.split(":", 1)
        files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        agg = {}
        for f in files:
            for row in csv.DictReader(open(f)):
                k = (short(row["Kernel_Name"]), row["Counter_Name"])
                a = agg.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
        for (kern, ctr), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{tag},{kern},{ctr},{n},{tot:.3f},{tot / n:.3f}")


if __name__ == "__main__":
    main()
