"""TEST INFRASTRUCTURE: order-independent seeded parameter fill, shared by ``oracle/make_golden.py`` (which applies it to the REAL
reference classes in the build container) and by the tests (which apply it to the oracle restatement and to the HIP model), so
fixtures need not store weights: every tensor of ``module.state_dict()`` is drawn from its own generator seeded by
``crc32(name) ^ seed``."""
import zlib

import torch


def fill_state_(module: torch.nn.Module, seed: int, scale: float = 1.0) -> torch.nn.Module:
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if not t.dtype.is_floating_point:
                continue                                   # num_batches_tracked
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "running_var":
                t.copy_(torch.rand(t.shape, generator=g) * 0.5 + 0.75)
            elif leaf == "running_mean":
                t.copy_(torch.randn(t.shape, generator=g) * 0.1)
            elif t.ndim <= 1:
                # norm scales / PReLU slopes / biases: keep scales near 1 and slopes near 0.25 so depth does not blow up
                parent = name.rsplit(".", 1)[0].rsplit(".", 1)[-1]
                if leaf == "bias":
                    t.copy_(torch.randn(t.shape, generator=g) * 0.05)
                elif "prelu" in parent:
                    t.copy_(torch.rand(t.shape, generator=g) * 0.3 + 0.1)
                else:
                    t.copy_(torch.rand(t.shape, generator=g) * 0.4 + 0.8)
            else:
                fan_in = t[0].numel()
                t.copy_(torch.randn(t.shape, generator=g) * (scale / fan_in ** 0.5))
    return module


def checksums(module: torch.nn.Module):
    return {k: (v.double().sum().item(), (v.double() ** 2).sum().item()) for k, v in module.state_dict().items() if v.dtype.is_floating_point}
