"""Time of nf_final_bwd (+ reduce) at the benchmark shape on random saved tensors; NF_MI355X_LIB selects an ablation build
(tools/build_variant.py fbX "-DFB_ABL_NOIDENT | -DFB_ABL_NOSPLINE | -DFB_ABL_NOROWS | -DFB_ABL_NOMFMA" final_bwd.hip: wrong results,
timing only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nfa
from normflows_amd import ops
from bench import build_c2_model
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
m = build_c2_model(num_layers=1).to(dev)
c = m.flows[0].prqct
u, net = c.unconditional_transform, c.transform_net
x = torch.randn(B, 64, device=dev)
assert c._fused_eligible(x, None)
gy, gld = torch.randn(B, 64, device=dev), torch.randn(B, device=dev)
cond24 = 0.5 * torch.randn(B, 32, 24, device=dev)
blob = c._train_blob_for(x)
_, wpad, _, wfull_t = c._train_buffers(x)
lin = [l for blk in net.blocks for l in blk.linear_layers]
d = lambda t: t.detach()
ops.rqs_fused_pack_all(blob, d(net.initial_layer.weight), d(net.initial_layer.bias), [d(l.weight) for l in lin], [d(l.bias) for l in lin],
                       d(net.final_layer.weight), d(net.final_layer.bias), d(u.unnormalized_widths), d(u.unnormalized_heights),
                       d(u.unnormalized_derivatives), wfull=wfull_t, wpad=wpad, identity_idx=c.identity_features)
f = lambda: ops.final_bwd(x, gy, gld, cond24, wpad, blob, d(u.unnormalized_widths), d(u.unnormalized_heights),
                          d(u.unnormalized_derivatives), c._fused_parity, len(net.blocks))
if "--trace" in sys.argv:      # build: tools/build_variant.py fb_TRACE "-DFB_TRACE" final_bwd.hip
    import ctypes
    tr = torch.zeros(16, dtype=torch.int64, device=dev)
    nfa._lib.lib().nf_final_bwd_debug_trace(ctypes.c_void_p(tr.data_ptr()))
    for _ in range(3): f()
    torch.cuda.synchronize()
    t = tr.cpu().tolist()
    names = ["tile prologue", "identity half", "wait parameter rows", "spline x2 + piece write", "stage wait + barrier", "row stores + requests",
             "A reads + MFMA issue", "epilogue"]
    tot = sum(t[:8])
    for n, v in zip(names, t):
        print("  %-26s %9d cycles  %5.1f %%" % (n, v, 100.0 * v / max(tot, 1)))
    print("  total %d cycles (workgroup 0, wave 0, all its tiles)" % tot)
    sys.exit(0)
for _ in range(5): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30): f()
e.record(); torch.cuda.synchronize()
print("%s: %.1f us per call (nf_final_bwd + nf_final_bwd_reduce + allocations), B = %d" % (
    os.environ.get("NF_MI355X_LIB", "product build").split("/")[-1], s.elapsed_time(e) / 30 * 1e3, B))
