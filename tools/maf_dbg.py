#!/usr/bin/env python3
"""Round-trip and run-to-run determinism of the one-pass MAF inverse (nf_maf_inverse) on the config-5 layer at growing batch
sizes: the check that caught a too-permissive counted vmcnt wait which every small-batch parity test passed.
NF_MI355X_LIB selects a build variant."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import normflows_amd as nfa
dev = torch.device("cuda:0")
torch.manual_seed(0)
f = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2).to(dev)
for B in (256, 4096, 16384, 65536):
    x = torch.randn(B, 128, device=dev)
    with torch.no_grad():
        outs = []
        for rep in range(3):
            z, ld = f.inverse(x)
            outs.append(z.clone())
        xr, _ = f.forward(outs[0])
    err = (xr - x).abs()
    bad_rows = (err.max(1).values > 1e-3).nonzero().flatten()
    print("B=%d roundtrip max %.2e  bad rows %d  rep-to-rep max diff %.2e" % (B, float(err.max()), bad_rows.numel(),
          float((outs[0] - outs[1]).abs().max())))
    if bad_rows.numel():
        r = int(bad_rows[0]); fe = (err[r] > 1e-3).nonzero().flatten()
        print("   first bad row %d (wave %d lane %d), first bad feature %d, n bad feats %d; bad rows mod 64 hist:" % (r, r // 64, r % 64, int(fe[0]), fe.numel()),
              torch.bincount(bad_rows % 64, minlength=64).tolist()[:8], "waves:", torch.unique(bad_rows // 64)[:10].tolist())
