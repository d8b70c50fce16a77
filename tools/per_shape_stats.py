#!/usr/bin/env python3
"""Per-shape kernel summary from a rocprofv3 kernel trace: launches of one kernel are grouped by grid size AND by the
HIP stream they ran on, so the actor's 6,400-row launches (actor stream) and the learner's 512-row launches (learner
stream) of the same kernel get separate rows -- `rocprofv3 --stats` mixes them in one average, and the persistent
kernels launch min(256, rows) blocks for every batch size, so the grid alone does not tell them apart.

  python tools/per_shape_stats.py gpurun_out/final/prof_bench profiles/r02_bench_kernel_per_shape.csv
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = name.replace("rela_amd::(anonymous namespace)::", "").replace("rela_amd::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main(root, out):
    path = max(glob.glob(os.path.join(root, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1),
               int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), int(r["Workgroup_Size_X"]), "stream %s" % r["Stream_Id"])
        groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in groups.values())
    rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "blocks_x", "grid_y", "grid_z", "threads", "stream", "calls", "avg_us", "min_us", "max_us",
                    "percent"])
        for (name, bx, gy, gz, th, cl), v in rows:
            if sum(v) < 0.0005 * total:
                continue
            w.writerow([name, bx, gy, gz, th, cl, len(v), "%.1f" % (sum(v) / len(v) / 1e3), "%.1f" % (min(v) / 1e3),
                        "%.1f" % (max(v) / 1e3), "%.2f" % (100.0 * sum(v) / total)])
    print("wrote", out, "from", path)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
