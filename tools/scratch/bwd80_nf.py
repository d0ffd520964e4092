import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import os, sys, torch
sys.path.insert(0, %r)
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, n, H, d = 16, int(sys.argv[1]), 8, int(sys.argv[2])
C = H * d
g = torch.Generator().manual_seed(7)
qkv = torch.randn(B * n, 3 * C, generator=g).half().cuda()
do = torch.randn(B * n, C, generator=g).half().cuda()
pre = Recorder(dev)
lse = torch.empty((B, H, n), dtype=torch.float32, device=dev)
o = pre.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=H, nq=n, nk=n, d=d, lse=lse)
pre.run()
rec = Recorder(dev)
rec.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, batch=B, heads=H, nq=n, nk=n, d=d)
for _ in range(300): rec.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): rec.run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 200 * 1e3
print("n=%%d d=%%d NF=%%s  %%.1f us  %%.3f of peak" %% (n, d, os.environ.get("PV_ATTN_BWD_NF"), us, 10.0 * B * H * n * n * d / us / 1e6 / 2500))
""" % ROOT
for n, d in ((1024, 80), (256, 160)):
    for nf in ("0", "1", "2", "3"):
        r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(d)], env=dict(os.environ, PV_ATTN_BWD_NF=nf), capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-400:], flush=True)
