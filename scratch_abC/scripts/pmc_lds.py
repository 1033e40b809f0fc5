"""LDS bank conflicts per kernel from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE pass:
conflict % = SQ_LDS_BANK_CONFLICT (extra cycles) / SQ_LDS_IDX_ACTIVE (all LDS-array cycles), MI355X_MICROARCH.md section LDS.
usage: python scripts/pmc_lds.py gpurun_out/pmc4/..._counter_collection.csv"""
import sys

from pmc_summary import load


def main():
    per, _ = load(sys.argv[1])
    rows = []
    for k, c in per.items():
        act, conf = sum(c.get("SQ_LDS_IDX_ACTIVE", [0])), sum(c.get("SQ_LDS_BANK_CONFLICT", [0]))
        if act > 0:
            rows.append((act, k, len(c["SQ_LDS_IDX_ACTIVE"]), conf))
    rows.sort(reverse=True)
    print("%-72s %5s %14s %14s %9s" % ("kernel", "n", "LDS_IDX_ACTIVE", "BANK_CONFLICT", "conflict%"))
    for act, k, n, conf in rows[:30]:
        print("%-72s %5d %14.3e %14.3e %8.1f%%" % (k[:72], n, act, conf, 100.0 * conf / act))


if __name__ == "__main__":
    main()
