"""Per-dispatch counter dump of a rocprofv3 --pmc CSV: one line per (kernel dispatch), counters side by side.
   python tools/pmc_dump.py <counter_collection.csv> [kernel substring]"""
import csv
import sys


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    by = {}
    names = {}
    for r in csv.DictReader(open(path)):
        if sub and sub not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        by.setdefault(d, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        names[d] = r["Kernel_Name"][:60]
    cols = sorted({c for v in by.values() for c in v})
    print("dispatch kernel " + " ".join(cols))
    for d in sorted(by):
        print(d, names[d], " ".join("%.4g" % by[d].get(c, float("nan")) for c in cols))


if __name__ == "__main__":
    main()
