#!/usr/bin/env python3
"""Idle time of the GPU inside the bench's timed steps, from a rocprofv3 --kernel-trace CSV: the union of all kernel
intervals against the span they cover, per window of STEPS steps' worth of `conv12_s3<false>` launches at 6,400 rows (two per
step), and the largest gaps with the kernels on either side.   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
# the f32x3 headline region: a stretch where conv12_s3<false> launches are dense
idx = [i for i, r in enumerate(rows) if "conv12_s3<false>" in r[2]]
if not idx:
    sys.exit("no conv12_s3<false> launches in the trace")
lo, hi = idx[len(idx) // 4], idx[len(idx) // 4 + 40]  # 40 launches from the first quarter on = ~20 steps
seg = rows[lo:hi]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
gaps = []
prev = seg[0]
for r in seg[1:]:
    if r[0] > cur_e:
        busy += cur_e - cur_s
        gaps.append((r[0] - cur_e, prev[2][:60], r[2][:60]))
        cur_s, cur_e = r[0], r[1]
    else:
        cur_e = max(cur_e, r[1])
    if r[1] >= prev[1]:
        prev = r
busy += cur_e - cur_s
span = t1 - t0
print("span %.3f ms, busy (union of kernels) %.3f ms = %.1f %%, %d kernels, %d gaps, idle %.3f ms" % (
    span / 1e6, busy / 1e6, 100.0 * busy / span, len(seg), len(gaps), (span - busy) / 1e6))
gaps.sort(reverse=True)
for g, a, b in gaps[:12]:
    print("  gap %7.1f us after %-60s before %s" % (g / 1e3, a, b))
tot = {}
for s, e, n, q in seg:
    tot[n[:70]] = tot.get(n[:70], 0) + (e - s)
print("sum of kernel durations %.3f ms (overlap factor %.2f)" % (sum(tot.values()) / 1e6, sum(tot.values()) / busy))
