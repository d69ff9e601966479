#!/usr/bin/env python3
"""Within-process A/B of rqs_fused.hip build variants (-D flags) on the GPU box: each variant is compiled into its own
shared object, packed weights are shared, and launches of all variants are interleaved round-robin (cdna guide 5.4
rule 24).  Usage: python tools/fused_ablate.py "" "-DNF_ABL_NOEPI" ...   (timing only; ablations are NOT correct)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from bench import build_c2_model, c2_inputs  # noqa: E402
from normflows_amd._lib import f64, i32, i64, ptr  # noqa: E402

CSRC = os.path.join(ROOT, "normalizing-flows_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "ablate")
os.makedirs(OUT, exist_ok=True)


def build(flags, idx):
    so = os.path.join(OUT, "v%d.so" % idx)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so,
           os.path.join(CSRC, "rqs_fused.hip"), os.path.join(CSRC, "rqs_fused_x3.hip")] + [f for f in flags.split() if f != "X3"]
    subprocess.check_call(cmd)
    lib = C.CDLL(so)
    return lib


DIRECTION = int(os.environ.get("NF_DIR", "0"))


def main():
    variants = sys.argv[1:] or [""]
    libs = [build(f, i) for i, f in enumerate(variants)]
    dev = torch.device("cuda:0")
    model = build_c2_model(num_layers=2).to(dev)
    x = c2_inputs().to(dev)
    crqs, lu = model.flows[0], model.flows[1]
    blob = crqs.prqct._fused_blob(lu)
    y = torch.empty_like(x)
    ld = torch.zeros(len(x), device=dev)
    st = nfa._lib.stream()

    from normflows_amd import ops
    blob3 = ops.rqs_fused_x3_pack(blob, 2, True)
    use_x3 = ["X3" in f.split() for f in variants]

    def launch(lib):
        if use_x3[libs.index(lib)]:
            rc = lib.nf_rqs_fused_x3(ptr(x), ptr(y), ptr(ld), ptr(blob3), i32(0), i32(1), i64(len(x)), i32(64), i32(128),
                                     i32(2), i32(8), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), i32(DIRECTION), i32(1), st)
        else:
            rc = lib.nf_rqs_fused(ptr(x), ptr(y), ptr(ld), ptr(blob), i32(0), i32(1), i64(len(x)), i32(64), i32(128),
                                  i32(2), i32(8), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), i32(DIRECTION), i32(1), st)
        assert rc == 0, rc

    for lib in libs:
        for _ in range(3):
            launch(lib)
    torch.cuda.synchronize()
    rounds, per = 10, 8
    res = [[] for _ in libs]
    for _ in range(rounds):
        for i, lib in enumerate(libs):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(per):
                launch(lib)
            e.record()
            torch.cuda.synchronize()
            res[i].append(s.elapsed_time(e) / per * 1e3)
    for f, r in zip(variants, res):
        r = sorted(r)
        print("%-50s median %.1f us  min %.1f us" % (f or "(baseline)", r[len(r) // 2], r[0]))


if __name__ == "__main__":
    main()
