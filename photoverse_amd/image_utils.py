"""Image pre / post-processing either side of the hot path, with the reference's geometry.

* ``clip_image_processor`` - what ``CLIPImageProcessor()(images=img)`` does in ``/root/reference/generate.py:57``
  ([EXT] transformers defaults): convert to RGB, resize the SHORT side to 224 (bicubic, aspect ratio kept,
  long side = int(224 * long / short)), centre crop 224x224 (top = (h - 224) // 2), rescale 1/255, normalise with the CLIP
  mean / std.
* ``preprocess_image`` - ``/root/reference/datasets/utils.py:139-157`` ([EXT] torchvision ``Resize(size)`` +
  ``CenterCrop(size)`` + ``ToTensor`` + ``Normalize(0.5, 0.5)``): short side to ``size`` with the chosen interpolation
  (long side = int(size * long / short)), centre crop (top = round((h - size) / 2)), [-1, 1].
* ``denormalize`` / ``to_pil`` - ``/root/reference/utils/image_utils.py:6-29``.

Host-side PIL / numpy work on single images (load time, not the hot path).
"""
from __future__ import annotations

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _pil_resample(interpolation: str):
    from PIL import Image
    return {"nearest": Image.NEAREST, "bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC, "lanczos": Image.LANCZOS}[interpolation]


def _resize_short_side(img, size: int, resample):
    """Short side -> ``size``, aspect ratio kept; the long side is truncated like torchvision / transformers do."""
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        nw, nh = size, int(size * h / w)
    else:
        nw, nh = int(size * w / h), size
    return img.resize((nw, nh), resample)


def clip_image_processor(img, size: int = 224) -> torch.Tensor:
    """PIL image -> (3, size, size) float32, CLIP-normalised."""
    if img.mode != "RGB":
        img = img.convert("RGB")
    img = _resize_short_side(img, size, _pil_resample("bicubic"))
    w, h = img.size
    top, left = (h - size) // 2, (w - size) // 2
    img = img.crop((left, top, left + size, top + size))
    arr = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() * (1.0 / 255.0)
    mean = torch.tensor(CLIP_MEAN)[:, None, None]
    std = torch.tensor(CLIP_STD)[:, None, None]
    return (arr - mean) / std


def preprocess_image(raw_image, size: int = 512, interpolation: str = "bicubic") -> torch.Tensor:
    """PIL image -> (3, size, size) float32 in [-1, 1] (the VAE-side pixels, ``datasets/utils.py:139-157``)."""
    img = raw_image if raw_image.mode == "RGB" else raw_image.convert("RGB")
    img = _resize_short_side(img, size, _pil_resample(interpolation))
    w, h = img.size
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    img = img.crop((left, top, left + size, top + size))
    arr = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
    return (arr - 0.5) / 0.5


def denormalize(image: torch.Tensor) -> torch.Tensor:
    """[-1, 1] -> [0, 1] (``utils/image_utils.py:6-11``)."""
    return (image / 2 + 0.5).clamp(0, 1)


def to_pil(image: torch.Tensor):
    """(3, H, W) in [0, 1] -> PIL (``utils/image_utils.py:24-30``)."""
    from PIL import Image
    arr = image.detach().float().cpu().permute(1, 2, 0).numpy()
    return Image.fromarray((arr * 255).round().astype("uint8"))


def denormalize_clip(image: torch.Tensor, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """CLIP-normalised pixels -> [0, 1] (``utils/image_utils.py:14-21``)."""
    m = torch.tensor(mean, dtype=image.dtype, device=image.device)[:, None, None]
    s = torch.tensor(std, dtype=image.dtype, device=image.device)[:, None, None]
    return (image * s + m).clamp(0, 1)


def save_images_grid(grid_data, img_grid_file, header: int = 50):
    """``grid_data``: [(column title, [PIL images])], one column per entry, one row per sample; a white header strip carries the titles
    (``{}`` in a title becomes ``S*``).  The layout of ``utils/image_utils.py:32-69`` (train.py:549-596 writes one per sample step)."""
    from PIL import Image, ImageDraw, ImageFont
    titles = [t for t, _ in grid_data]
    columns = [imgs for _, imgs in grid_data]
    rows = [np.concatenate([np.asarray(im.convert("RGB")) for im in row], axis=1) for row in zip(*columns)]
    arr = np.pad(np.concatenate(rows, axis=0), ((header, 0), (0, 0), (0, 0)), mode="constant", constant_values=255)
    grid = Image.fromarray(arr.astype("uint8"), "RGB")
    draw = ImageDraw.Draw(grid)
    try:
        font = ImageFont.truetype("arial.ttf", 36)
    except IOError:
        try:
            font = ImageFont.load_default(size=36)
        except TypeError:                                   # older Pillow: fixed-size bitmap font
            font = ImageFont.load_default()
    width = columns[0][0].width
    for i, text in enumerate(titles):
        text = text.format("S*")
        x1, y1, x2, y2 = font.getbbox(text)
        draw.text(((width - (x2 - x1)) // 2 + i * width, (header - (y2 - y1)) // 2), text, font=font, fill="black")
    grid.save(img_grid_file)
    return grid
