#!/usr/bin/env python3
"""Linear layers whose launch has fewer than 256 tiles of 256 x 320 (the N = 1280 layers of the 16 x 16 level in the merged plan, M = 8192; the N = 640 layers
of the 32 x 32 level per branch, M = 16384): the 128 x 160 kernel (pv_gemm.hip) against the 128-row form of the 8-wave tile (PV_GEMM_BIG128=1: 128 x 320, one
workgroup per CU).  One child process per form (the switch is read once), sustained timing, results compared.

usage (GPU box): python tools/diag/linear128_ab.py [rounds]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, torch, hashlib
sys.path.insert(0, %r)
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
torch.manual_seed(0)
for (M, K, N, res, big_min) in ((8192, 1280, 1280, True, 256), (8192, 5120, 1280, True, 256), (8192, 1280, 1280, False, 256), (16384, 640, 640, True, 256),
                               (16384, 2560, 640, True, 256), (16384, 1280, 640, False, 256), (4096, 1280, 1280, True, 128), (4096, 5120, 1280, True, 128)):
    x = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * K ** -0.5).half()
    r = torch.randn(M, N, device=dev).half() if res else None
    rec = Recorder(dev)
    rec.big_min = big_min
    out = rec.gemm(x, w, bias=torch.zeros(N, device=dev), residual=r)
    for _ in range(400): rec.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): rec.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    ref = x.float() @ w.float().t() + (r.float() if res else 0)
    err = ((out.float() - ref).norm() / ref.norm()).item()
    print("M=%%d K=%%d N=%%d res=%%d min=%%d  %%-48s wgs=%%-5d %%7.1f us  %%6.0f TFLOP/s  err %%.1e  md5 %%s" %% (M, K, N, res, big_min, rec.tags[0][0], rec.tags[0][3], us, 2.0 * M * K * N / us / 1e6, err,
          hashlib.md5(out.cpu().numpy().tobytes()).hexdigest()[:8]), flush=True)
""" % ROOT
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for r in range(rounds):
    for env in ({"PV_GEMM_BIG128": "0"}, {"PV_GEMM_BIG128": "640"}):
        out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
        print("round %d %s" % (r, env), flush=True)
        print(out.stdout + out.stderr[-400:], flush=True)
