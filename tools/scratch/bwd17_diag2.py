import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, H, N, d = 1, 2, 2048, 40
VAR = sys.argv[1]
C = H * d
g = torch.Generator().manual_seed(3)
qkv = torch.randn(B * N, 3 * C, generator=g).half().cuda()
do_full = torch.randn(B * N, C, generator=g).half().cuda()
os.environ["PV_ATTN8_BWD_MIN"] = "1"
recs = {}
do = torch.zeros_like(do_full)
for var in ("1", VAR):
    os.environ["PV_ATTN8_BWD"] = var
    rec = Recorder(dev)
    lse = rec.empty((B, H, N), torch.float32)
    o = rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=H, nq=N, nk=N, d=d, lse=lse)
    outs = rec.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, batch=B, heads=H, nq=N, nk=N, d=d)
    recs[var] = (rec, outs)
line = []
for s in list(range(4)) + [N // 32 - 1]:
    do.zero_()
    do[32 * s:32 * s + 32] = do_full[32 * s:32 * s + 32]
    res = {}
    for var in ("1", VAR):
        rec, outs = recs[var]
        rec.run(); torch.cuda.synchronize()
        res[var] = [t.float().clone() for t in outs]
    e = [((a - b).norm() / (b.norm() + 1e-20)).item() for a, b in zip(res[VAR], res["1"])]
    line.append("s%02d dq %.1e dk %.1e dv %.1e" % (s, *e))
print("\n".join(line))
