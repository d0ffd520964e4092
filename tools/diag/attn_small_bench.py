import os, sys, torch
sys.path.insert(0, os.getcwd())
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
def t(d, n, B=16, H=8):
    C = H * d
    qkv = torch.randn(B * n, 3 * C, device=dev).half()
    rec = Recorder(dev)
    rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=H, nq=n, nk=n, d=d)
    for _ in range(500): rec.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): rec.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print("d=%d n=%d: %.1f us (%.3f of peak) %s" % (d, n, us, 4.0 * B * H * n * n * d / us / 1e6 / 2500, rec.tags[-1][0]), flush=True)
t(80, 1024); t(160, 256); t(40, 4096)
