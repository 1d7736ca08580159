"""rocprofv3 --kernel-trace CSV -> per (kernel, grid) statistics.   python tests/tools/kernel_stats_by_grid.py <kernel_trace.csv> [top]

`--stats` averages a kernel over EVERY launch of the process; bench.py's legs launch the same instantiation on different grids (the 4-image
slot plans `value` is timed on, the one-image plan of the `bs1` figure, the 10-image plans of `config.alt_issue`), so the average the
roofline is checked against is the one of the timed plan's GRID: this table splits them."""
import collections
import csv
import sys


def grid_of(row):
    if "Grid_Size" in row:
        return int(row["Grid_Size"])
    g = 1
    for ax in "XYZ":
        g *= int(row.get(f"Grid_Size_{ax}", 1) or 1)
    return g


def main(path, top=24):
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].replace("void ", "").split("(")[0]
        dur[(name, grid_of(row))].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    total = sum(sum(v) for v in dur.values())
    print('"Name","GridThreads","Calls","TotalDurationUs","AverageUs","Percentage","MinUs","MaxUs"')
    for (name, grid), v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print(f'"{name}",{grid},{len(v)},{sum(v):.1f},{sum(v) / len(v):.2f},{100 * sum(v) / total:.2f},{min(v):.2f},{max(v):.2f}')


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24)
