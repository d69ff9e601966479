import ctypes as C, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import normflows_amd as nfa
from normflows_amd import ops
lib = nfa._lib.lib()
dev = "cuda:0"
B = 65536
x = torch.randn(B, 128, device=dev); W1 = torch.randn(128, 128, device=dev); W2 = torch.randn(128, 128, device=dev); b = torch.randn(128, device=dev)
t = torch.randn(B, 128, device=dev)
buf = torch.zeros(8, dtype=torch.int64, device=dev)
for name, fn in (("forward", lambda: ops.rows_block(x, W1, b, W2, b)), ("backward", lambda: ops.rows_block(x, W2, None, W1, None, trans=True, mask1=t, mask2=x, relu=False))):
    for _ in range(3): fn()
    lib.nf_rows_block_debug_trace(C.c_void_p(buf.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.nf_rows_block_debug_trace(C.c_void_p(0))
    v = buf.cpu().double() * 0.01
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    print("%s: panels %.1f us, x load + product 1 + epilogue 1 %.1f us, product 2 + epilogue 2 %.1f us, workgroup total %.1f us; launch-to-launch %.1f us" % (
        name, v[1]-v[0], v[3]-v[1], v[4]-v[3], v[4]-v[0], s.elapsed_time(e)/20*1e3))
