import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, H, N, d = 2, 2, 2048, 40
C = H * d
g = torch.Generator().manual_seed(3)
qkv = torch.randn(B * N, 3 * C, generator=g).half().cuda()
do = torch.randn(B * N, C, generator=g).half().cuda()
os.environ["PV_ATTN8_BWD_MIN"] = "1"
outs = {}
for var in ("1", "17", "17"):
    os.environ["PV_ATTN8_BWD"] = var
    rec = Recorder(dev)
    lse = rec.empty((B, H, N), torch.float32)
    o = rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=H, nq=N, nk=N, d=d, lse=lse)
    dq, dk, dv = rec.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, batch=B, heads=H, nq=N, nk=N, d=d)
    rec.run(); torch.cuda.synchronize()
    res = [t.float().clone() for t in (dq, dk, dv)]
    if var == "1":
        ref = res
        continue
    for name, a, b in zip(("dq", "dk", "dv"), res, ref):
        e = (a - b).view(B, N // 64, 64, H, d)
        r = b.view(B, N // 64, 64, H, d)
        blk = (e.pow(2).sum((2, 4)) / r.pow(2).sum((2, 4))).sqrt()       # [B][block][H]
        print(var, name, "total %.3e" % ((a - b).norm() / b.norm()).item())
        for bb in range(B):
            for hh in range(H):
                print("   b%d h%d " % (bb, hh) + " ".join("%.0e" % x for x in blk[bb, :, hh].tolist()))
