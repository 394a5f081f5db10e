"""Turns the two rocprofv3 PMC passes over tools/microbench_resblock.py into profiles/<name>_traffic.json.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d A -o r -- python3 tools/microbench_resblock.py --store bf16 --reps 2
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d B -o r -- python3 tools/microbench_resblock.py --store bf16 --reps 2
  python tools/pmc_traffic.py A/*counter_collection.csv B/*counter_collection.csv profiles/r01_v17_pmc_resblock_traffic.json

FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads
(/opt/skills/guides/MI355X_MICROARCH.md), so it is doubled.  Launches are matched to the micro-benchmark's loop order
(C, (taps, dil), activation), each configuration being launched 1 + reps times; the values of a configuration are averaged."""
import csv
import json
import sys

ORDER = [(C, mult, k, dil, act) for C, mult in ((256, 8), (128, 48), (64, 192), (32, 384))
         for k, dil in ((3, 1), (7, 3), (11, 5)) for act in ("lrelu", "snake")]


def per_config(path, counter, per):
    rows = [r for r in csv.DictReader(open(path)) if "resblock_step_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    assert len(rows) == per * len(ORDER), (len(rows), per, len(ORDER))
    return [sum(float(r["Counter_Value"]) for r in rows[i * per:(i + 1) * per]) / per for i in range(len(ORDER))]


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    batch, frames = 32, 640
    fetch = per_config(fetch_csv, "FETCH_SIZE", 1 + reps)
    write = per_config(write_csv, "WRITE_SIZE", 1 + reps)
    launches = []
    for (C, mult, k, dil, act), f, w in zip(ORDER, fetch, write):
        rows = batch * frames * mult
        launches.append({"kernel": f"resblock_step<{C}>", "C": C, "taps": k, "dil": dil, "act": act, "rows": rows, "store": "bf16",
                         "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_corrected": int(2 * f * 1024 + w * 1024),
                         "algorithmic_bytes": rows * C * 2 * 2})
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on tools/microbench_resblock.py "
            "--store bf16 --reps %d; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads; the "
            "2-byte snake loads are uncalibrated, so the snake rows are an upper bound)" % reps)
    json.dump({"note": note, "launches": launches}, open(out, "w"), indent=1)
    for r in launches:
        print(r["kernel"], r["taps"], r["act"], "%.2f x algorithmic" % (r["hbm_bytes_corrected"] / r["algorithmic_bytes"]))


if __name__ == "__main__":
    main()
