#!/usr/bin/env python3
"""Turns the rocprofv3 PMC passes of tools/pmc_passes.sh (pmc_table.json) plus the kernel-trace stats of a bench.py run into
profiles/r04_pmc_counters.json, the file bench.py's `roofline` object reads hardware-counter figures from.

    tools/make_pmc_profile.py <pmc dir with pmc_table.json> <kernel_stats.csv of the same workload> [out.json]

Per pass (ssao = depth_pairs_kernel + ssao_kernel, blur = blur_pair_kernel + blur_replay_chain_kernel, light) and per launch
of the 4K bench frame:
  hbm_bytes_per_launch  (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
                        (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact
  valu_issue_frac       SQ_ACTIVE_INST_VALU * 4 cycles (the SQ_ACTIVE_INST_* counters tick in quad-cycles, MI355X_MICROARCH.md) /
                        (1024 SIMDs * kernel cycles), the cycles MEASURED in the very pass that counted the instructions:
                        SQ_BUSY_CYCLES / 32 (rocprofv3 sums the 32 shader engines).  Round 2 assumed 2.4 GHz over the unprofiled
                        duration and got fractions above 1; profiled dispatches run near 2.0 GHz and are not to be mixed with
                        unprofiled timings.  GRBM_GUI_ACTIVE / 8 is recorded beside it (reads high on dispatches this short)
  l2_read_GBs           TCP_TCC_READ_REQ_sum * 64 B / kernel time
  ta_busy_frac          TA_TA_BUSY_sum / 256 TAs / kernel cycles
  bound                 the largest of the fractions, by name -- a label, not a diagnosis: round 4 cut light_kernel's VALU instructions by 21 %
                        without moving its duration; what bounds it is (wavefronts x exposed latency) / resident wavefronts over a
                        streaming floor (profiles/r04_experiments.txt, occupancy sweep)
The file records the sha of the kernel sources it was taken on (bench.kernel_source_hash); bench.py prints null instead of
these figures when the sources have changed since."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIMDS, TAS = 1024, 256


def main():
    pmc_dir, stats_csv = sys.argv[1], sys.argv[2]
    out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r04_pmc_counters.json")
    table = json.load(open(os.path.join(pmc_dir, "pmc_table.json")))
    dur = {}
    for r in csv.DictReader(l for l in open(stats_csv) if not l.startswith("#")):
        if "cry::" in r["Name"]:
            dur[r["Name"].split("(")[0].replace("void ", "")] = (float(r["AverageNs"]) * 1e-3, int(r["Calls"]))

    def find(d, key):
        return [v for k, v in d.items() if key in k]

    def agg(keys_weights):
        """sum over kernels (name fragment, launches per frame) of per-launch counters and durations"""
        tot = {"us": 0.0}
        for frag, n in keys_weights:
            cs = find(table, frag)
            ds = find(dur, frag)
            if not cs or not ds:
                raise SystemExit("kernel %r missing from the PMC table or the stats file" % frag)
            tot["us"] += n * ds[0][0]
            for c, v in cs[0].items():
                tot[c] = tot.get(c, 0.0) + n * v
        return tot

    import bench
    args = type("A", (), {})()
    wl = {"width": 3840, "height": 2160, "blur_count": 4, "lights": 3, "pcf": "literal", "shadow_dim": 4096, "camera": "reference"}
    passes = {"ssao": [("depth_pairs_kernel", 1), ("ssao_kernel<true, true, true, false>", 1)],
              "blur": [("blur_pair_kernel<true>", 1), ("blur_replay_chain_kernel", 1)],
              "light": [("light_kernel<true, false, false>", 1)]}
    kernels = {}
    for name, kw in passes.items():
        t = agg(kw)
        cyc = t.get("SQ_BUSY_CYCLES", 0.0) / 32.0           # summed over the pass's launches, like every other counter
        if cyc <= 0.0:
            raise SystemExit("SQ_BUSY_CYCLES missing from the PMC table: the fractions need measured cycles")
        fr = {"valu": t.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (SIMDS * cyc), "ta": t.get("TA_TA_BUSY_sum", 0.0) / TAS / cyc}
        hbm = (2.0 * t.get("FETCH_SIZE", 0.0) + t.get("WRITE_SIZE", 0.0)) * 1024.0
        fr["hbm"] = hbm / (t["us"] * 1e-6) / 8.0e12
        kernels[name] = {"us_per_frame": round(t["us"], 1), "hbm_bytes_per_launch": int(hbm),
                         "valu_insts": int(t.get("SQ_INSTS_VALU", 0)), "valu_issue_frac": round(fr["valu"], 3),
                         "ta_busy_frac": round(fr["ta"], 3), "hbm_frac_by_counters": round(fr["hbm"], 3),
                         "l2_read_GBs": round(t.get("TCP_TCC_READ_REQ_sum", 0.0) * 64.0 / (t["us"] * 1e-6) / 1e9, 1),
                         "l2_hit_rate": round(t.get("TCC_HIT_sum", 0.0) / max(1.0, t.get("TCC_HIT_sum", 0.0) + t.get("TCC_MISS_sum", 0.0)), 3),
                         "kernel_cycles": int(cyc), "grbm_gui_active_over_8": int(t.get("GRBM_GUI_ACTIVE", 0.0) / 8.0),
                         "bound": max(fr, key=fr.get)}
    out = {"workload": wl, "kernel_source_hash": bench.kernel_source_hash(),
           "cycles": "SQ_BUSY_CYCLES / 32 of the profiled dispatches (measured in the pass that counted the instructions; no clock assumed)",
           "method": "rocprofv3 --pmc, one pass per counter group, tools/pmc_passes.sh on the torch-free tools/prof_driver; durations from "
                     "rocprofv3 --kernel-trace --stats of bench.py; hbm = (2*FETCH_SIZE + WRITE_SIZE) KiB",
           "kernels": kernels, "per_kernel_counters": {k: v for k, v in table.items() if "cry::" in k}}
    json.dump(out, open(out_path, "w"), indent=1)
    for k, v in kernels.items():
        print(k, v)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
