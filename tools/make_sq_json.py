#!/usr/bin/env python3
"""profiles/r0N_sq.json from separate rocprofv3 --pmc passes (tools/pmc_passes.sh) over tools/pmc_probe.py: SQ wave-instruction
counts per launch of every profiling slot, keyed like bench.py's kernel slots, with the hash of the device code they belong to.

  tools/make_sq_json.py PMC_DIR_PREFIX OUT.json [batch width height nfeatures [config]]      (PMC_DIR_PREFIX/p1 .. p4)
  tools/make_sq_json.py PMC_DIR_PREFIX OUT.json [batch width height nfeatures [config]]      (PMC_DIR_PREFIX/p1 .. p5)
Per slot it also carries the clock the chip HELD during the kernel (GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's own duration in the
same pass: MI355X_MICROARCH.md, DVFS note) and the static rate-class mix of its code (tools/valu_mix.py).  bench.py's
roofline.ports = per-port busy fractions of the dominant kernel over its LIVE launch duration at that clock."""
import csv, collections, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from orb_slam2_detailed_comments_amd import build

SLOTS = {"k_pyr_l0": ["k_pyr_l0"], "k_pyr_resize": ["k_pyr_resize_rows", "k_pyr_resize"], "k_fast_rows": ["k_fast_rows"],
         "k_quadtree": ["k_quadtree"], "k_describe": ["k_describe"], "k_match": ["k_match", "k_match_f4", "k_match_merge", "k_stereo_rows", "k_stereo_batch", "k_stereo_cut"]}


def main():
    prefix, out = sys.argv[1:3]
    B, W, H, NF = (int(a) for a in sys.argv[3:7]) if len(sys.argv) >= 7 else (256, 640, 480, 1000)
    config = sys.argv[7] if len(sys.argv) >= 8 else "tum"     # bench.py --config name the counters were taken on
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(lambda: collections.defaultdict(list))   # dispatch durations (ns) of the pass a counter was taken in
    for fn in glob.glob(prefix + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            name = r["Kernel_Name"].split("(")[0].split("<")[0]
            name = name[5:] if name.startswith("void ") else name
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[name][r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    import valu_mix
    mix = valu_mix.mix(build.LIB)
    steps = len(agg.get("k_describe", {}).get("SQ_INSTS_VALU", [])) or 1
    kernels = {}
    for slot, names in SLOTS.items():
        cs = collections.defaultdict(float)
        launches = 0
        for n in names:
            for c, v in agg.get(n, {}).items():
                cs[c] += sum(v)
            if n != "k_match_merge":
                launches += len(agg.get(n, {}).get("SQ_INSTS_VALU", []))
        if not launches:
            continue
        kernels[slot] = {c: int(v / launches) for c, v in sorted(cs.items())}   # per launch of the slot
        kernels[slot]["launches_per_step"] = launches / steps
        gui = [(v, d) for n in names for v, d in zip(agg.get(n, {}).get("GRBM_GUI_ACTIVE", []), dur.get(n, {}).get("GRBM_GUI_ACTIVE", []))]
        if gui:   # cycles summed over the 8 XCDs / dispatch duration (reads high on dispatches under ~0.3 ms: the guide's caveat)
            kernels[slot]["clock_ghz"] = round(sum(v for v, _ in gui) / 8.0 / sum(d for _, d in gui), 3)
            kernels[slot]["clock_pass_avg_launch_us"] = round(sum(d for _, d in gui) / len(gui) / 1e3, 2)
        main = next((n for n in names if n in mix and agg.get(n)), None)
        if main:
            kernels[slot]["valu_mix_static"] = mix[main]
    json.dump({"config": config, "kernels_sha256_16": build.kernels_hash(), "batch": B, "width": W, "height": H, "nfeatures": NF,
               "steps_profiled": steps, "unit": "wave-instructions (or the counter's own unit) per launch", "kernels": kernels},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
