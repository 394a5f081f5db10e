"""Summarises the SQ counter pass over tools/microbench_resblock.py (see tools/collect_profiles.sh) per configuration.

  python tools/pmc_sq_summary.py profiles/r01_v17_pmc_resblock_SQ.csv profiles/r01_v17_pmc_resblock_SQ_summary.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (32 per v_mfma_f32_32x32x16_bf16); SQ_BUSY_CYCLES is summed over the
32 shader engines.  Derived per launch:
  simd_cycles   = SQ_BUSY_CYCLES / 32 * 1024          (kernel duration in cycles x 1024 SIMDs)
  mfma_util     = SQ_VALU_MFMA_BUSY_CYCLES / simd_cycles
  waves_per_simd= 4 * SQ_WAVE_CYCLES / simd_cycles
  active / wait_any / wait_inst = share of wave lifetime issuing / parked at s_waitcnt or a barrier / stalled at issue
  valu_per_elem = 64 * SQ_INSTS_VALU / (rows * C)     (vector lane-operations per tensor element)"""
import csv
import json
import sys

from pmc_traffic import ORDER


def main():
    src, out = sys.argv[1:3]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    per = 1 + reps
    rows = [r for r in csv.DictReader(open(src)) if "resblock_step_kernel" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [by[k] for k in sorted(by)]
    assert len(disp) == per * len(ORDER), (len(disp), per, len(ORDER))
    res = []
    for i, (C, mult, k, dil, act) in enumerate(ORDER):
        grp = disp[i * per:(i + 1) * per]
        c = {n: sum(g[n] for g in grp) / per for n in grp[0]}
        simd = c["SQ_BUSY_CYCLES"] / 32 * 1024
        wave = c["SQ_WAVE_CYCLES"]
        elems = 32 * 640 * mult * C
        res.append({"kernel": f"resblock_step<{C}>", "taps": k, "dil": dil, "act": act,
                    "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4), "waves_per_simd": round(4 * wave / simd, 2),
                    "active": round(c["SQ_ACTIVE_INST_ANY"] / wave, 3), "wait_any": round(c["SQ_WAIT_ANY"] / wave, 3),
                    "wait_inst": round(c["SQ_WAIT_INST_ANY"] / wave, 3), "valu_share_of_active": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_ACTIVE_INST_ANY"], 3),
                    "valu_per_elem": round(64 * c["SQ_INSTS_VALU"] / elems, 1), "counters": {n: round(v) for n, v in c.items()}})
    json.dump({"note": __doc__, "launches": res}, open(out, "w"), indent=1)
    for r in res:
        print("%-18s k=%-2d %-5s mfma_util %.3f waves/simd %.2f active %.2f wait_any %.2f wait_inst %.2f valu/elem %.0f" % (
            r["kernel"], r["taps"], r["act"], r["mfma_util"], r["waves_per_simd"], r["active"], r["wait_any"], r["wait_inst"], r["valu_per_elem"]))


if __name__ == "__main__":
    main()
