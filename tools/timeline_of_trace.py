#!/usr/bin/env python3
"""Timeline of the LAST burst of kernels in a rocprofv3 --kernel-trace CSV (the last library call of tools/upload_probe.py): consecutive
launches of one kernel are merged; offsets in ms from the first kernel of the burst.  Usage: tools/timeline_of_trace.py <kernel_trace.csv> [gap_ms]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 30e6
# bursts = library calls: each scan starts with k_colgemv (v = S a_hat) after its arena memset; cut at the last `k_cert_accumulate`
ends = [i for i, r in enumerate(rows) if r[2].startswith("k_cert_accumulate")]
print("%d kernels, %d scans in the trace" % (len(rows), len(ends)))
start = ends[-2] + 1 if len(ends) > 1 else 0
burst = rows[start:ends[-1] + 1] if ends else rows
t0 = burst[0][0]
out = []
for s, e, nme in burst:
    if out and out[-1][2] == nme and s - out[-1][1] < 2e6:
        out[-1] = (out[-1][0], max(e, out[-1][1]), nme, out[-1][3] + 1, out[-1][4] + (e - s))
    else:
        out.append((s, e, nme, 1, e - s))
for s, e, nme, cnt, busy in out:
    print("%8.2f .. %8.2f ms  %-60s x%-4d busy %7.2f ms" % ((s - t0) / 1e6, (e - t0) / 1e6, nme, cnt, busy / 1e6))
