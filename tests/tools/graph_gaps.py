"""Dev tool: where a one-image forward's time goes INSIDE a hipGraph replay -- kernel durations and the gaps between consecutive kernels, from a
rocprofv3 --kernel-trace of `bench.py --pipeline 0` (one graph replay per image, the host waiting for each).

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --pipeline 0 --steps 20 --warmup 5 --no-cpu-baseline --repeat-blocks 0
    python tests/tools/graph_gaps.py DIR > profiles/rNN_graph_gaps_b1.txt
"""
import csv
import glob
import sys


def main(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dd3d::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    # a forward starts at the fused stem kernel and ends at nms_finalize; take the forwards of the timed block (the last ones before the probe)
    starts = [i for i, k in enumerate(ks) if k[0].startswith("stem_fused")]
    fwd = []
    for a, b in zip(starts, starts[1:]):
        seg = ks[a:b]
        if any(k[0].startswith("nms_finalize") for k in seg):
            end = max(i for i, k in enumerate(seg) if k[0].startswith("nms_finalize"))
            fwd.append(seg[:end + 1])
    n = max(set(len(x) for x in fwd), key=[len(x) for x in fwd].count)
    fwd = [x for x in fwd if len(x) == n][-20:]
    print(f"{len(fwd)} forwards of {n} kernels each (the last of the run)")
    tot_d = tot_g = 0.0
    for i in range(n):
        dur = sorted((x[i][2] - x[i][1]) / 1e3 for x in fwd)
        gap = sorted(((x[i][1] - x[i - 1][2]) / 1e3 if i else 0.0) for x in fwd)
        md, mg = dur[len(dur) // 2], gap[len(gap) // 2]
        tot_d += md
        tot_g += mg
        print(f"{i:3d} {fwd[0][i][0][:70]:70s} duration {md:8.2f} us   gap before {mg:7.2f} us")
    span = sorted((x[-1][2] - x[0][1]) / 1e3 for x in fwd)
    print(f"sum of median durations {tot_d:.1f} us + gaps {tot_g:.1f} us; first kernel start -> last kernel end, median {span[len(span) // 2]:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
