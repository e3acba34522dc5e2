#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection.csv files into one JSON summary per kernel.

usage: pmc_summary.py OUT.json KERNEL_SUBSTRING CSV [CSV ...]

For every counter found in the CSVs the summary holds the mean value per launch over the launches
of kernels whose name contains KERNEL_SUBSTRING.  When both FETCH_SIZE and WRITE_SIZE are present
(they have to come from separate passes: together they exceed the TCC counter slots) it adds
`hbm_bytes_per_launch` = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB: both counters are in KiB, and on
gfx950 FETCH_SIZE tallies 128-byte fabric read requests as 64 bytes, so wide coalesced reads are
doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    out, needle, files = sys.argv[1], sys.argv[2], sys.argv[3:]
    sums, counts, names = defaultdict(float), defaultdict(int), set()
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if needle not in row["Kernel_Name"]:
                    continue
                names.add(row["Kernel_Name"].split("(")[0])
                sums[row["Counter_Name"]] += float(row["Counter_Value"])
                counts[row["Counter_Name"]] += 1
    mean = {k: sums[k] / counts[k] for k in sums}
    summary = {"kernel": sorted(names), "launches": dict(counts), "mean_per_launch": mean,
               "sources": [f.split("gpurun_out/")[-1] for f in files]}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        summary["fetch_bytes_corrected"] = 2 * 1024 * mean["FETCH_SIZE"]
        summary["write_bytes"] = 1024 * mean["WRITE_SIZE"]
        summary["hbm_bytes_per_launch"] = summary["fetch_bytes_corrected"] + summary["write_bytes"]
    with open(out, "w") as fh:
        json.dump(summary, fh, indent=1)
        fh.write("\n")
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
