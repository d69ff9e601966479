#!/usr/bin/env python3
"""Per-GlowBlock time inside the level launches of config 4 from a rocprofv3 --kernel-trace of tools/config_bench.py 4.

    python tools/glow_level_chains.py <kernel_trace.csv> [--json profiles/rNN_config4_glow_level_chains.json]

A level chain = the 32 GlowBlocks of a level in ONE nf_glow_level launch (config_bench runs log_prob with the chains, and
once per block for comparison).  Launches longer than 8 x the kernel's shortest one are chains.  FLOP per block:
2 (9 Cin 256 + 256^2 + 256 9 Cout) per pixel x 256 images (the conditioner only)."""
import argparse
import collections
import csv
import json
import statistics

LEVELS = {  # kernel-name fragment -> (label, Cin, Cout, pixels per image)
    "glow_convnet_kernel": ("16x16", 6, 12, 256),
    "glow_convnet_small_kernel": ("8x8", 12, 24, 64),
    "glow_convnet_tiny_kernel": ("4x4", 24, 48, 16),
}
PEAK = 157.3e12   # fp32 MFMA, MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--json")
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(a.trace)):
        for frag in LEVELS:
            if frag + "<" in r["Kernel_Name"] or frag + "(" in r["Kernel_Name"] or r["Kernel_Name"].endswith(frag):
                dur[frag].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    res = {"what": "config 4 (Glow L=3, K=32, B=%d): durations of the nf_glow_level launches from rocprofv3 --kernel-trace of "
                   "tools/config_bench.py 4; a level chain = %d GlowBlocks in one launch" % (a.batch, a.blocks), "kernels": {}}
    for frag, d in dur.items():
        label, cin, cout, px = LEVELS[frag]
        lo = min(d)
        chains = [x for x in d if x > 8 * lo]
        singles = [x for x in d if x <= 8 * lo]
        flop = 2.0 * (9 * cin * 256 + 256 * 256 + 256 * 9 * cout) * px * a.batch
        e = {"level": label, "conditioner_gflop_per_block": flop / 1e9}
        if chains:
            per = statistics.median(chains) / a.blocks
            e.update(level_chain_launches=len(chains), level_chain_median_us=statistics.median(chains), per_block_us_in_chain=per,
                     frac_of_fp32_mfma_peak=flop / (per * 1e-6) / PEAK)
        if singles:
            e.update(single_block_launches=len(singles), single_block_median_us=statistics.median(singles))
        res["kernels"][frag] = e
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.json:
        open(a.json, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
