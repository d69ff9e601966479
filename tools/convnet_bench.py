"""GlowBlock conditioner: nf_glow_convnet against the library path (MIOpen / hipBLASLt convolutions + bias-activation
passes) at the three levels of BASELINE config 4 (B = 256).  usage: python tools/convnet_bench.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for cin, cout, H, W in ((6, 12, 16, 16), (12, 24, 8, 8), (24, 48, 4, 4)):
    torch.manual_seed(0)
    net = nfa.nets.ConvNet2d([cin, 256, 256, cout], [3, 1, 3], 0.0, init_zeros=False).to(dev)
    x = torch.randn(256, cin, H, W, device=dev)
    c1, _, c2, _, c3 = net.net
    layout = nfa.ops.glow_convnet_layout(256, H, W)
    blob = nfa.ops.glow_convnet_pack(*[p.detach() for p in (c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias)],
                                     layout=layout)
    flop = 2.0 * 256 * H * W * (cin * 9 * 256 + 256 * 256 + 256 * 9 * cout)
    with torch.no_grad():
        tl = timed(lambda: net._forward_inference(x))
        tf = timed(lambda: nfa.ops.glow_convnet(x, blob, cout, 0.0, layout))
        err = float((net._forward_inference(x) - nfa.ops.glow_convnet(x, blob, cout, 0.0, layout)).abs().max())
    print("ConvNet2d [%d,256,256,%d] on 256x%dx%d: library %.1f us (%.1f TFLOP/s)  one launch (layout %d) %.1f us (%.1f TFLOP/s, "
          "%.2f of 157.3)  max|diff| %.1e" % (cin, cout, H, W, tl * 1e6, flop / tl / 1e12, layout, tf * 1e6, flop / tf / 1e12,
                                      flop / tf / 157.3e12, err))
