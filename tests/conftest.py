import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


class _LazyWeights:
    """Full-size seeded state dicts (``oracle.fullsize`` builders: construction of default-initialised modules only, no oracle
    arithmetic), built on first use and shared by every test of the session."""

    def __init__(self):
        self._cache = {}

    def __call__(self, name):
        if name not in self._cache:
            from oracle import fullsize as fs
            build = {"unet": fs.unet_state, "text": fs.text_state, "vision": fs.vision_state, "image_adapter": lambda: fs.adapter_state(3),
                     "text_adapter": lambda: fs.adapter_state(4), "vae": fs.vae_state}[name]
            self._cache[name] = build()
        return self._cache[name]


@pytest.fixture(scope="session")
def full_weights():
    return _LazyWeights()


@pytest.fixture(scope="session")
def full_hip_unet(full_weights):
    """The SD-v1.5-sized HIP UNet (859.5 M parameters + PhotoVerse processors) with the session's seeded weights, on the GPU."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from oracle import fullsize as fs
    from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
    with fs.no_init():
        hip = UNet2DConditionModel()
        set_visual_cross_attention_adapter(hip, (5,))
    hip.load_state_dict(full_weights("unet"))
    hip.to("cuda")
    return hip
