"""Run a few full-size training iterations (bench.py's train_step leg alone) - the program to put after `rocprofv3 --kernel-trace --stats --`."""
import json
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402
from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    unet = UNet2DConditionModel()
    set_visual_cross_attention_adapter(unet, (5,))
    unet.to(dev)
    print(json.dumps(bench.train_step_leg(unet, 16, 64, dev, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 2)))
