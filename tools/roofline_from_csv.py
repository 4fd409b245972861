#!/usr/bin/env python3
"""Recompute a bench line's `roofline.frac_in_step` from rocprofv3's kernel_stats.csv of the same command.

    python tools/roofline_from_csv.py profiles/r5/bench_train.json profiles/r5/bench_train_kernel_stats.csv --csv-steps N

The bench line carries the dominant family's algorithmic work per step (`roofline.algorithmic_work_per_step`) and the device
kernels the family launches (`roofline.csv_kernels`); the csv carries each kernel's total duration over the profiled run.  N = the
number of training steps that run executed (bench.py runs 3 initialisation steps + the calibration + the settle groups + warm-up + the
timed steps + two per-op passes; with --brief only the first four), printed by `bench.py` as `steps_executed` - or pass --per-step to
take the family's csv total divided by its calls x the bench line's launches per step."""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bench_json")
    ap.add_argument("kernel_stats_csv")
    ap.add_argument("--csv-steps", type=float, default=None, help="training steps the profiled run executed")
    a = ap.parse_args()
    line = [ln for ln in open(a.bench_json) if ln.startswith("{")][-1]
    r = json.loads(line)["roofline"]
    names = r["csv_kernels"]
    tot_ns, calls, rows = 0.0, 0, []
    for row in csv.DictReader(open(a.kernel_stats_csv)):
        k = row["Name"]
        base = k.split("(")[0]
        if any(base.endswith(n) or ("::" + n + "<") in k or ("::" + n + "(") in k or k.startswith("_ZN12_GLOBAL__N_1%d%s" % (len(n), n)) for n in names):
            tot_ns += float(row["TotalDurationNs"])
            calls += int(row["Calls"])
            rows.append((k[:70], int(row["Calls"]), float(row["TotalDurationNs"]) / 1e6))
    for k, c, ms in rows:
        print("  %-70s calls %6d  total %9.3f ms" % (k, c, ms))
    steps = a.csv_steps
    if steps is None:
        raise SystemExit("pass --csv-steps (the bench line of the profiled run prints steps_executed)")
    work = r["algorithmic_work_per_step"] * steps
    div = 1e9 if r["bound"] == "hbm" else 1e12
    ach = work / (tot_ns * 1e-9) / div
    print("family %s: %.3f ms per step over %g steps -> %.1f %s = %.4f of the %.0f peak (bench line: frac_in_step %s, frac alone %s)"
          % (r["kernel"], tot_ns / 1e6 / steps, steps, ach, r["unit"], ach / r["peak"], r["peak"], r.get("frac_in_step"), r["frac"]))


if __name__ == "__main__":
    main()
