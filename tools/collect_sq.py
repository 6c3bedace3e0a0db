#!/usr/bin/env python3
"""Summarise rocprofv3 SQ / GRBM counter passes (one directory per pass, --output-format csv) into a JSON of
per-LAUNCH averages per kernel, with the derived figures DESIGN.md quotes:

    python tools/collect_sq.py OUT.json DIR [DIR ...]

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Derived per kernel:
    gpu_cycles            = GRBM_GUI_ACTIVE / 8                       (duration of one launch in shader cycles)
    valu_busy             = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * gpu_cycles)
    waves_per_simd        = 4 * SQ_WAVE_CYCLES / (1024 * gpu_cycles)
    issue_2cyc_frac       = 2 * SQ_INSTS_VALU / (1024 * gpu_cycles)    (the guide's full-rate VALU issue as the floor)
"""
import collections
import csv
import glob
import json
import sys

KERNELS = ["k_fast_cells_wave", "k_quadtree", "k_blur", "k_describe", "k_match_knn", "k_match_init", "k_resize", "k_copy_level0",
           "k_ba_iteration", "k_ba_classify", "k_pose_opt", "k_pyramid", "k_ba_ph_chi2", "k_ba_ph_hll", "k_ba_ph_hpp", "k_ba_ph_prep",
           "k_ba_ph_dinv", "k_ba_ph_schur2", "k_ba_ph_chol", "k_ba_ph_chol_band", "k_ba_ph_fillplan", "k_ba_ph_decide", "k_ba_ph_round", "k_ba_ph_backsub", "k_ba_ph_trial", "k_ba_ph_accept", "k_ba_ph_finish"]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    for d in dirs:
        for f in glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
                if name == "k_match_knn_mfma":  # phase 1 of the matcher on the matrix cores: same stage as k_match_knn
                    name = "k_match_knn"
                if name not in KERNELS:
                    continue
                c = r["Counter_Name"]
                agg[name][c] += float(r["Counter_Value"])
                disp[name][c].add((f, r["Dispatch_Id"]))
    res = {"_note": "per-launch averages; SQ_* cycle counters are quad-cycles summed over waves, GRBM over 8 XCDs"}
    for k in KERNELS:
        if k not in agg:
            continue
        a = {c: v / max(len(disp[k][c]), 1) for c, v in agg[k].items()}
        e = {c: round(v, 1) for c, v in a.items()}
        e["launches_sampled"] = max(len(s) for s in disp[k].values())
        if "GRBM_GUI_ACTIVE" in a:
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0
            e["gpu_cycles"] = round(cyc, 1)
            simd_cycles = 1024.0 * cyc
            if "SQ_ACTIVE_INST_VALU" in a:
                e["valu_busy"] = round(4 * a["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4)
            if "SQ_WAVE_CYCLES" in a:
                e["waves_per_simd"] = round(4 * a["SQ_WAVE_CYCLES"] / simd_cycles, 3)
            if "SQ_INSTS_VALU" in a:
                e["issue_2cyc_frac"] = round(2 * a["SQ_INSTS_VALU"] / simd_cycles, 4)
            if "SQ_ACTIVE_INST_LDS" in a:
                e["lds_pipe_busy_per_cu"] = round(4 * a["SQ_ACTIVE_INST_LDS"] / (256.0 * cyc), 4)
        res[k] = e
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k != "_note":
            print(k, {c: v[c] for c in ("gpu_cycles", "valu_busy", "issue_2cyc_frac", "waves_per_simd", "lds_pipe_busy_per_cu") if c in v})


if __name__ == "__main__":
    main()
