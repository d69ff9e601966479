"""Phase timestamps (100 MHz wall clock) of workgroup 0 / wave 0 of nf_resblock_bwd built with -DNF_BB_TRACE."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nfa
from normflows_amd import ops, _lib as L
dev = torch.device('cuda:0')
B = 65536
gh = torch.randn(B, 128, device=dev); t = torch.randn(B, 128, device=dev); h = torch.randn(B, 128, device=dev)
W1 = torch.randn(128, 128, device=dev) * 0.1; W2 = torch.randn(128, 128, device=dev) * 0.1
x = torch.randn(B, 64, device=dev); wfull = torch.randn(128, 64, device=dev) * 0.1; wfull_t = wfull.t().contiguous(); gx = torch.zeros(B, 64, device=dev)
buf = torch.zeros(96, dtype=torch.int64, device=dev)
L.lib().nf_resblock_bwd_debug_trace(ctypes.c_void_p(buf.data_ptr()))
names = ["top", "dgrad1+epi", "wgrad2", "barrier M", "dgrad2+fold", "wgrad1", "barrier E", "(pre-loop)", "gh_in->Dt", "barrier Y", "wgrad3", "dgrad3+gx"]
for init in (False, True):
    for _ in range(3):
        buf.zero_()
        if init: ops.resblock_bwd(gh, t, h, W1, W2, x=x, wfull=wfull_t, gx=gx)
        else: ops.resblock_bwd(gh, t, h, W1, W2)
        torch.cuda.synchronize()
    v = buf.cpu().tolist()
    t0 = v[7]
    print("init", init, "| shader clock during the tile loop: %.0f MHz (clock64 ticks per 100 MHz wall-clock tick x 100)"
          % (100.0 * (v[5] - v[6]) / max(v[4] - v[7], 1)))
    for tc in range(1, 6):
        row = v[tc * 12: tc * 12 + 12]
        print("  tile %d: " % tc + "  ".join("%s %.2f" % (names[i], (row[i] - t0) / 100.0) for i in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11) if row[i]))
