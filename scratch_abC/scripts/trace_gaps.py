"""Idle time between kernels, from a rocprofv3 --kernel-trace CSV:  python scripts/trace_gaps.py <kernel_trace.csv> [steps]

Takes the last `steps` (default 2) training steps of the trace (a step = the span between two launches of the optimizer
kernel), and prints per step: wall span, the union of kernel-busy intervals over all queues, idle time, the number of
launches, and a histogram of the idle gaps -- what a hipGraph capture or further fusion could at most recover."""
import csv
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:44]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    marks = [i for i, r in enumerate(rows) if "adam" in r[2].lower()]
    if len(marks) < steps + 1:
        print("not enough optimizer launches in the trace: %d" % len(marks))
        return
    for s in range(steps):
        lo, hi = marks[-steps - 1 + s] + 1, marks[-steps + s] + 1
        seg = rows[lo:hi]
        t0, t1 = seg[0][0], max(r[1] for r in seg)
        busy, cur_s, cur_e, gaps = 0, seg[0][0], seg[0][1], []
        for a, b, _, _ in seg[1:]:
            if a > cur_e:
                busy += cur_e - cur_s
                gaps.append(a - cur_e)
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        busy += cur_e - cur_s
        span = t1 - t0
        queues = len(set(r[3] for r in seg))
        hist = [0, 0, 0, 0, 0]
        for g in gaps:
            hist[0 if g < 1000 else 1 if g < 2000 else 2 if g < 4000 else 3 if g < 8000 else 4] += 1
        print("step %d: %d launches on %d queues, span %.3f ms, busy (union) %.3f ms, idle %.3f ms in %d gaps "
              "(<1us %d, 1-2us %d, 2-4us %d, 4-8us %d, >8us %d; total of the >4us gaps %.3f ms)"
              % (s, len(seg), queues, span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps), *hist, sum(g for g in gaps if g >= 4000) / 1e6))
    # every gap of the first of those steps, in time order, with the kernels (and queues) either side
    seg = rows[marks[-steps - 1] + 1:marks[-steps] + 1]
    cur_e, cur_name, cur_q = seg[0][1], seg[0][2], seg[0][3]
    t0 = seg[0][0]
    for a, b, name, q in seg[1:]:
        if a > cur_e:
            print("   t=%8.1f us  gap %5.1f us  after [q%s] %-44s before [q%s] %s" % ((a - t0) / 1e3, (a - cur_e) / 1e3, cur_q, short(cur_name), q, short(name)))
        if b > cur_e:
            cur_e, cur_name, cur_q = b, name, q


if __name__ == "__main__":
    main()
