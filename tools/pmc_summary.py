#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic per launch.

Correction per MI355X_MICROARCH.md §HBM: counters are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced streaming read (doubled here); WRITE_SIZE reads bytes exactly for 16-B-per-lane streaming stores and float atomics.
Usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(?:\d+)?([a-z][a-z0-9_]*_kernel)", name)
    if not m or m.group(1).startswith("__amd"):
        return None
    k = m.group(1)
    if "EpiHeads" in name:
        k += "<EpiHeads>"
    e = re.search(r"EpiDenseS<(\d+)u>", name)      # compile-time epilogue mode: tells FFN1 (19) from FFN2 (1) from the vocab GEMMs (0)
    if e:
        k += "<mode%s>" % e.group(1)
    return k


def load(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        acc[(k, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for key in sorted(set(fetch) | set(write)):
        f = fetch.get(key, [0.0])
        w = write.get(key, [0.0])
        fb = 2.0 * 1024.0 * sum(f) / len(f)
        wb = 1024.0 * sum(w) / len(w)
        out["%s|grid=%d" % key] = dict(launches=len(f), fetch_bytes=round(fb), write_bytes=round(wb), hbm_bytes=round(fb + wb))
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:25]:
        print("%-55s n=%4d fetch=%8.1f MB write=%8.1f MB" % (k, v["launches"], v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
