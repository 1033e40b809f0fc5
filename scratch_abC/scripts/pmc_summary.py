"""Per-kernel summary of the rocprofv3 --pmc passes written by scripts/gpu_round.sh (phase "pmc"):
   pass 1: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES     pass 3: FETCH_SIZE     pass 4: WRITE_SIZE
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration;
HBM bytes per launch = FETCH_SIZE * 2 (gfx950: FETCH_SIZE counts 128-byte requests as 64, MI355X_MICROARCH.md) * 1 KiB... the
counter unit is KiB (rocprofv3 derived metric), WRITE_SIZE likewise (uncalibrated for narrow stores).
usage: python scripts/pmc_summary.py gpurun_out/pmc1/lbc_counter_collection.csv [pmc3.csv] [pmc4.csv]
"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z0-9_]+?)I(.*)E+v", name)
    if m:
        return m.group(1) + "<" + m.group(2)[:40] + ">"
    return name[:70]


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values (per dispatch)
    dur = collections.defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per, dur


def main():
    p1, d1 = load(sys.argv[1])
    fetch = load(sys.argv[2])[0] if len(sys.argv) > 2 else {}
    write = load(sys.argv[3])[0] if len(sys.argv) > 3 else {}
    rows = []
    for k, c in p1.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        gui = sum(c["GRBM_GUI_ACTIVE"])
        mfma = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))
        n = len(c["GRBM_GUI_ACTIVE"])
        t = sum(d1[k].values())
        busy = mfma / (gui / 8.0 * 1024.0) if gui else 0.0
        clk = gui / 8.0 / t if t else 0.0                       # cycles per ns = GHz
        fk = sum(fetch.get(k, {}).get("FETCH_SIZE", [0])) / max(1, len(fetch.get(k, {}).get("FETCH_SIZE", [1])))
        wk = sum(write.get(k, {}).get("WRITE_SIZE", [0])) / max(1, len(write.get(k, {}).get("WRITE_SIZE", [1])))
        rows.append((t, k, n, t / n / 1e3, busy, clk, fk * 2 * 1024 / 1e6, wk * 1024 / 1e6))
    rows.sort(reverse=True)
    print("%-64s %5s %9s %9s %6s %10s %10s" % ("kernel", "n", "avg us", "MFMAbusy", "GHz", "fetch MB", "write MB"))
    for t, k, n, avg, busy, clk, fm, wm in rows[:28]:
        print("%-64s %5d %9.1f %9.3f %6.2f %10.1f %10.1f" % (k[:64], n, avg, busy, clk, fm, wm))


if __name__ == "__main__":
    main()
