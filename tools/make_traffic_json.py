#!/usr/bin/env python3
"""profiles/r0N_traffic.json from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_probe.py.

  tools/make_traffic_json.py FETCH_DIR WRITE_DIR OUT.json [batch width height nfeatures [config]]

HBM bytes per launch = FETCH_SIZE[KB] * 1024 * 2 (gfx950 calibration, tools/fetch_calib.hip) + WRITE_SIZE[KB] * 1024.
Entries are keyed by the profiling slot bench.py uses (k_pyr_l0, k_pyr_resize, k_fast_rows, ...); the device kernels
behind a slot are summed (k_match = k_match + k_match_merge) and listed."""
import csv, collections, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_detailed_comments_amd import build

SLOTS = {
    "k_pyr_l0": ["k_pyr_l0", "k_pyr_l0_color"],
    "k_pyr_resize": ["k_pyr_resize", "k_pyr_resize_rows", "k_pyr_resize_flat"],
    "k_fast_rows": ["k_fast_rows"],
    "k_quadtree": ["k_quadtree"],
    "k_describe": ["k_describe"],
    "k_match": ["k_match", "k_match_f4", "k_match_merge", "k_stereo_rows", "k_stereo_batch", "k_stereo_cut"],   # (stereo configs: the stereo match)
}


def collect(path, counter):
    agg = collections.defaultdict(list)
    for fn in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].split("<")[0]           # template instances: "void k_describe<0>(...)"
                agg[name[5:] if name.startswith("void ") else name].append(float(r["Counter_Value"]))
    return agg


def main():
    fdir, wdir, out = sys.argv[1:4]
    B, W, H, NF = (int(a) for a in sys.argv[4:8]) if len(sys.argv) >= 8 else (256, 640, 480, 1000)
    config = sys.argv[8] if len(sys.argv) >= 9 else "tum"     # bench.py --config name the counters were taken on
    fe, wr = collect(fdir, "FETCH_SIZE"), collect(wdir, "WRITE_SIZE")
    steps = len(fe.get("k_describe", [])) or 1
    kernels = {}
    for slot, names in SLOTS.items():
        f_kb = sum(sum(fe.get(n, [])) for n in names) / steps      # per step
        w_kb = sum(sum(wr.get(n, [])) for n in names) / steps
        launches = sum(len(fe.get(n, [])) for n in names if n != "k_match_merge") / steps
        if launches == 0:
            continue
        kernels[slot] = {
            "device_kernels": [n for n in names if n in fe],
            "launches_per_step": launches,
            "FETCH_SIZE_KB_per_step": round(f_kb, 1), "WRITE_SIZE_KB_per_step": round(w_kb, 1),
            "hbm_bytes_per_launch": int((f_kb * 2.0 + w_kb) * 1024 / launches),
        }
    total = sum(k["hbm_bytes_per_launch"] * k["launches_per_step"] for k in kernels.values())
    json.dump({"config": config, "kernels_sha256_16": build.kernels_hash(), "batch": B, "width": W, "height": H, "nfeatures": NF, "steps_profiled": steps,
               "calibration": {"FETCH_SIZE_factor": 2.0, "WRITE_SIZE_factor": 1.0,
                               "how": "tools/fetch_calib.hip streams 512 MiB once with 4 B/lane, 16 B/lane and 64-byte row "
                                      "segments: FETCH_SIZE reports exactly 1/2 for all three, WRITE_SIZE is exact"},
               "hbm_bytes_per_step": int(total), "hbm_bytes_per_frame": int(total / B), "kernels": kernels},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
