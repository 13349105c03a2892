"""Prints a rocprofv3 kernel_stats.csv as a short table (durations per call and per N batches)."""
import csv
import sys

path, nb = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("%-46s calls %5s avg %9.1f us  per-batch %9.1f us  %6s%%" % (
        n, r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / nb / 1e3, r["Percentage"]))
