#!/usr/bin/env python3
"""Round-5 ablations of csrc/maf_inverse_h.hip on BASELINE configs[4]'s inverse pass (tools/maf_inverse_bench.py): every variant is
another build of the SAME source with -D switches (tools/build_variant.py -> lib/variants/maf5_<name>.so), run in its own process
with NF_MI355X_LIB.  The NO_* variants skip work and produce garbage: they are timing probes (what is the pass made of?), the
others (hnw8, lb4, nopair) are valid configurations.

    python tools/maf_ablate5.py build     # here (cross-compiles); the .so files travel with gpurun
    python tools/maf_ablate5.py run       # on the GPU box: one JSON line per variant
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {
    "base": "",
    "noseq": "-DNF_MAF_ABL_NO_SEQ",
    "nomfma": "-DNF_MAF_ABL_NO_MFMA",
    "nodma": "-DNF_MAF_ABL_NO_DMA",
    "nodma_noseq": "-DNF_MAF_ABL_NO_DMA -DNF_MAF_ABL_NO_SEQ",
    "nomfma_noseq": "-DNF_MAF_ABL_NO_MFMA -DNF_MAF_ABL_NO_SEQ",
    "hnw8": "-DNF_MAF_HNW=8",
    "hnw2": "-DNF_MAF_HNW=2",
    "lb8": "-DNF_MAF_LB=8",
    "nopair": "-DNF_MAF_ABL_NO_PAIR",
    # what the 3 ms between "MFMA issue alone" (5.8 ms) and nodma_noseq (8.9 ms) are made of: the same build with more removed
    "core": "-DNF_MAF_ABL_NO_DMA -DNF_MAF_ABL_NO_SEQ",
    "core_nobarrier": "-DNF_MAF_ABL_NO_DMA -DNF_MAF_ABL_NO_SEQ -DNF_MAF_ABL_NO_BARRIER",
    "core_nostage": "-DNF_MAF_ABL_NO_DMA -DNF_MAF_ABL_NO_SEQ -DNF_MAF_ABL_NO_STAGE",
    "core_nopublish": "-DNF_MAF_ABL_NO_DMA -DNF_MAF_ABL_NO_SEQ -DNF_MAF_ABL_NO_PUBLISH",
    "core_nobias": "-DNF_MAF_ABL_NO_DMA -DNF_MAF_ABL_NO_SEQ -DNF_MAF_ABL_NO_BIAS",
    "core_all": "-DNF_MAF_ABL_NO_DMA -DNF_MAF_ABL_NO_SEQ -DNF_MAF_ABL_NO_BARRIER -DNF_MAF_ABL_NO_STAGE -DNF_MAF_ABL_NO_PUBLISH -DNF_MAF_ABL_NO_BIAS",
    "nobarrier": "-DNF_MAF_ABL_NO_BARRIER",
    "nopublish": "-DNF_MAF_ABL_NO_PUBLISH",
    "puball": "-DNF_MAF_ABL_PUBLISH_ALL",      # also the tiles nobody streams back (the build before the last change of round 5)
}


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "run"
    names = sys.argv[2:] or list(VARIANTS)
    vdir = os.path.join(ROOT, "normalizing-flows_amd", "lib", "variants")
    if what == "build":
        for n in names:
            if VARIANTS[n]:
                subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "build_variant.py"), "maf5_" + n, VARIANTS[n],
                                       "maf_inverse_h.hip"])
        return
    for n in names:
        env = dict(os.environ)
        if VARIANTS[n]:
            env["NF_MI355X_LIB"] = os.path.join(vdir, "maf5_%s.so" % n)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "maf_inverse_bench.py"), "--reps", "4"], env=env,
                             capture_output=True, text=True, timeout=170)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if line:
            d = json.loads(line[0])
            print(json.dumps({"variant": n, "flags": VARIANTS[n], "inverse_pass_ms": round(d["inverse_pass_ms"], 3),
                              "round_trip_max_abs_err": d["round_trip_max_abs_err"]}), flush=True)
        else:
            print(json.dumps({"variant": n, "error": (out.stderr or out.stdout)[-300:]}), flush=True)


if __name__ == "__main__":
    main()
