#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes:
they do not fit one TCC pass) into profiles/traffic.json: HBM-side bytes per LAUNCH for every stage of the
front-end, which bench.py copies into roofline.traffic for the dominant stage.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json <batch>

Units/corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly HALF of the bytes of a wide coalesced streaming read (TCC_EA0_RDREQ tallies 128-B requests at 64 B),
so the read side is doubled; WRITE_SIZE is exact for streaming stores.  Byte-granular gathers (FAST staging,
describe) are outside the guide's calibrated patterns: their corrected figure is an upper bound.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import stage_source_hashes  # noqa: E402  (what bench.py compares the stamp with)

STAGE = {"k_ba_iteration": "ba_iteration", "k_pyramid": "pyramid", "k_copy_level0": "pyramid", "k_resize": "pyramid", "k_fast_cells": "fast", "k_fast_cells_wave": "fast", "k_quadtree": "quadtree",
         "k_blur": "blur", "k_describe": "describe", "k_match_knn": "match_knn", "k_match_init": "match_resolve"}
LAUNCHES_PER_STEP = {"pyramid": 1, "fast": 1, "quadtree": 1, "blur": 1, "describe": 1, "match_knn": 1, "match_resolve": 1,
                     "ba_iteration": 1}
PHASE = {"FETCH_SIZE": collections.defaultdict(float), "WRITE_SIZE": collections.defaultdict(float)}
BA_BATCH = 256  # only the one-workgroup-per-window batch launches of k_ba_iteration are tallied (grid = BA_BATCH x 512 threads)


def load(d, counter):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    tot = collections.defaultdict(float)
    calls = collections.Counter()
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    # The profiled command is the whole bench line, whose one-stream legs launch the SAME front-end kernels on single
    # frames: only the batch launches are tallied (a dispatch counts when its grid is within 1/32 of the largest grid of
    # its kernel -- the smallest pyramid level of a batch is still 40 x the largest level of one frame, a one-frame quadtree
    # launch 1/64 of a batch's).  Until late
    # round 3 the single-frame launches were counted as launches too, which diluted every per-launch figure of the
    # front-end by the share of those launches (x 0.78 with 46 tracked frames next to 160 batch launches).
    gmax = collections.defaultdict(int)

    def kname(r):
        return r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    for r in rows:
        gmax[kname(r)] = max(gmax[kname(r)], int(r["Grid_Size"]))
    for r in rows:
        name = kname(r)
        if name in STAGE and not name.startswith("k_ba") and int(r["Grid_Size"]) * 32 < gmax[name]:
            continue
        if name == "k_match_knn_mfma":  # phase 1 of the matcher on the matrix cores: same stage as k_match_knn
            name = "k_match_knn"
        if name == "k_ba_iteration" and int(r["Grid_Size"]) != BA_BATCH * 512:
            continue
        if name.startswith("k_ba_ph_"):  # the batch path as phase kernels: everything between two k_ba_ph_round launches is one round
            tot["ba_iteration"] += float(r["Counter_Value"])
            PHASE[counter][name] += float(r["Counter_Value"])
            if name == "k_ba_ph_round":
                calls["ba_iteration"] += 1
            continue
        if name in STAGE:
            tot[STAGE[name]] += float(r["Counter_Value"])
            if name in ("k_ba_iteration", "k_pyramid", "k_copy_level0", "k_fast_cells", "k_fast_cells_wave", "k_quadtree", "k_blur", "k_describe", "k_match_knn", "k_match_init"):
                calls[STAGE[name]] += 1
    return tot, calls


def main():
    fetch_dir, write_dir, out, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    fetch, calls = load(fetch_dir, "FETCH_SIZE")
    write, calls_w = load(write_dir, "WRITE_SIZE")
    res = {"_note": "HBM-side bytes per launch (one launch = one batch of %d frames; ba_iteration: one optimize() round of "
                    "%d windows, the average of the 5- and the 10-iteration round); fetch doubled per the gfx950 "
                    "FETCH_SIZE correction, write as counted" % (batch, BA_BATCH), "_batch": batch, "_ba_batch": BA_BATCH}
    if len(sys.argv) > 5:  # optional: SQ summary (tools/collect_sq.py) -> VALU instructions per launch for the issue-rate view
        sq = json.load(open(sys.argv[5]))
        for kern, st in (("k_fast_cells_wave", "fast"), ("k_quadtree", "quadtree"), ("k_blur", "blur"), ("k_describe", "describe"),
                         ("k_match_knn", "match_knn"), ("k_match_init", "match_resolve")):
            if kern in sq and "SQ_INSTS_VALU" in sq[kern]:
                res["valu_insts_" + st] = int(sq[kern]["SQ_INSTS_VALU"])
    for st in LAUNCHES_PER_STEP:
        if calls[st] == 0:
            continue
        f = fetch[st] / calls[st] * 1024.0 * 2.0
        w = write[st] / max(calls_w[st], 1) * 1024.0
        res[st] = int(f + w)
        res["_" + st + "_fetch_raw_kib"] = round(fetch[st] / calls[st], 1)
        res["_" + st + "_write_kib"] = round(write[st] / max(calls_w[st], 1), 1)
    if calls["ba_iteration"] and PHASE["FETCH_SIZE"]:  # bytes per optimize() round and phase kernel
        res["_ba_phase_bytes"] = {k: int((PHASE["FETCH_SIZE"][k] * 2.0 + PHASE["WRITE_SIZE"].get(k, 0.0)) * 1024.0 / calls["ba_iteration"])
                                  for k in sorted(PHASE["FETCH_SIZE"])}
    # stamp: bench.py nulls every stage whose kernel source no longer hashes to what the PMC passes ran
    res["_src_sha16"] = stage_source_hashes()
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        head = ""
    res["_collected"] = "%s%s" % (time.strftime("%Y-%m-%d"), (" after " + head) if head else "")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
