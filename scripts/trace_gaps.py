"""Idle share of a rocprofv3 kernel trace: sum of kernel durations vs the span they cover, and the gap histogram."""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[len(rows) // 3:]            # skip warm-up / initialisation
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
gaps.sort()
print("kernels %d  span %.3f ms  busy %.3f ms (%.1f%%)  median gap %.2f us  p90 gap %.2f us  mean gap %.2f us"
      % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, gaps[len(gaps) // 2] / 1e3, gaps[len(gaps) * 9 // 10] / 1e3,
         sum(gaps) / len(gaps) / 1e3))
