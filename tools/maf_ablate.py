#!/usr/bin/env python3
"""Within-process A/B of maf_inverse.hip build variants (-D flags) on one config-5 layer (d=128, hidden 512, B=65536).
Usage: python tools/maf_ablate.py "" "-DNF_MAF_ABL_NOSEQ" ...   (timing only; ablations are NOT correct)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from normflows_amd._lib import i32, i64, ptr  # noqa: E402

CSRC = os.path.join(ROOT, "normalizing-flows_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "ablate")
os.makedirs(OUT, exist_ok=True)


def build(flags, idx):
    so = os.path.join(OUT, "maf%d.so" % idx)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-o", so, os.path.join(CSRC, "maf_inverse.hip")] + flags.split())
    lib = C.CDLL(so)
    lib.nf_maf_inverse_scratch_floats.restype = C.c_int64
    return lib


def main():
    variants = sys.argv[1:] or [""]
    libs = [build(f, i) for i, f in enumerate(variants)]
    dev = torch.device("cuda:0")
    B, D, H = int(os.environ.get("NF_B", 65536)), 128, 512
    torch.manual_seed(0)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=2).to(dev)
    nfa.config.set_maf_tri(False)                  # round 2's kernel (maf_inverse.hip) reads format-0 packs only
    blob, table, Hp = layer._packed(dev)[:3]
    z = torch.randn(B, D, device=dev)
    y = torch.empty_like(z)
    ld = torch.zeros(B, device=dev)
    n = libs[0].nf_maf_inverse_scratch_floats(i64(B), i32(D), i32(Hp))
    scratch = torch.empty(n, device=dev)
    st = nfa._lib.stream()

    def launch(lib):
        rc = lib.nf_maf_inverse(ptr(z), ptr(y), ptr(ld), ptr(blob), ptr(table), ptr(scratch), i64(B), i32(D), i32(Hp), i32(1), st)
        assert rc == 0, rc

    for lib in libs:
        for _ in range(2):
            launch(lib)
    torch.cuda.synchronize()
    res = [[] for _ in libs]
    for _ in range(6):
        for i, lib in enumerate(libs):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(4):
                launch(lib)
            e.record()
            torch.cuda.synchronize()
            res[i].append(s.elapsed_time(e) / 4 * 1e3)
    for f, r in zip(variants, res):
        r = sorted(r)
        print("%-50s median %.1f us  min %.1f us" % (f or "(baseline)", r[len(r) // 2], r[0]))


if __name__ == "__main__":
    main()
