"""Host-side image preparation (CPU, PIL) — the reference does the same work on the CPU:

  dynamic_preprocess / find_closest_aspect_ratio   vita/util/data_utils_video_audio_neg_patch.py:1197-1255
  expand2square / process_images                   vita/model/language_model/vita_mixtral.py:384-415
  CLIP-style normalisation                         web_demo/vllm_tools/model_weight_file/preprocessor_config.json

Written against the behaviour, not the text, of those functions."""
import numpy as np
import torch
from PIL import Image

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def candidate_grids(min_num, max_num):
    grids = {(i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1) for j in range(1, n + 1)
             if min_num <= i * j <= max_num}
    return sorted(grids, key=lambda g: g[0] * g[1])


def find_closest_aspect_ratio(aspect_ratio, target_ratios, width, height, image_size):
    best, best_diff = (1, 1), float("inf")
    area = width * height
    for r in target_ratios:
        diff = abs(aspect_ratio - r[0] / r[1])
        if diff < best_diff:
            best, best_diff = r, diff
        elif diff == best_diff and area > 0.5 * image_size * image_size * r[0] * r[1]:
            best = r  # on a tie prefer the larger grid when the image has the pixels for it
    return best


def dynamic_preprocess(image, min_num=1, max_num=12, image_size=448, use_thumbnail=False):
    """Tile an image into a (cols x rows) grid of image_size squares closest to its aspect ratio;
    append a whole-image thumbnail when more than one tile.  Returns (tiles, [n_tiles])."""
    w, h = image.size
    cols, rows = find_closest_aspect_ratio(w / h, candidate_grids(min_num, max_num), w, h, image_size)
    resized = image.resize((image_size * cols, image_size * rows))
    tiles = []
    for i in range(cols * rows):
        x0, y0 = (i % cols) * image_size, (i // cols) * image_size
        tiles.append(resized.crop((x0, y0, x0 + image_size, y0 + image_size)))
    assert len(tiles) == cols * rows
    if use_thumbnail and len(tiles) != 1:
        tiles.append(image.resize((image_size, image_size)))
    return tiles, [len(tiles)]


def expand2square(pil_img, background_color):
    w, h = pil_img.size
    if w == h:
        return pil_img
    side = max(w, h)
    canvas = Image.new(pil_img.mode, (side, side), background_color)
    canvas.paste(pil_img, ((side - w) // 2, (side - h) // 2))
    return canvas


class ClipLikeImageProcessor:
    """resize (shortest edge, bicubic) -> centre crop -> /255 -> normalise; the subset of HF
    CLIPImageProcessor the demo touches: .preprocess(img, return_tensors='pt')['pixel_values'],
    __call__(list_of_images, return_tensors='pt'), .image_mean, .crop_size, .size."""

    def __init__(self, size=448, image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD):
        self.size = {"shortest_edge": size}
        self.crop_size = {"height": size, "width": size}
        self.image_mean, self.image_std = list(image_mean), list(image_std)
        self._s = size

    def _one(self, img):
        img = img.convert("RGB")
        w, h = img.size
        s = self._s
        if min(w, h) != s:
            if w <= h:
                nw, nh = s, int(s * h / w)
            else:
                nw, nh = int(s * w / h), s
            img = img.resize((nw, nh), resample=Image.BICUBIC)
            w, h = nw, nh
        left, top = (w - s) // 2, (h - s) // 2
        img = img.crop((left, top, left + s, top + s))
        a = np.asarray(img, dtype=np.float32) / 255.0
        a = (a - np.asarray(self.image_mean, np.float32)) / np.asarray(self.image_std, np.float32)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))

    def preprocess(self, images, return_tensors="pt", **_):
        if not isinstance(images, (list, tuple)):
            images = [images]
        return {"pixel_values": torch.stack([self._one(im) for im in images], 0)}

    __call__ = preprocess


def make_image_processor(size=448):
    return ClipLikeImageProcessor(size=size)


def process_images(images, image_processor, image_aspect_ratio=None):
    """vita_mixtral.py:397-415: optional pad-to-square with the mean colour, then the processor."""
    if image_aspect_ratio == "pad":
        out = []
        for im in images:
            im = expand2square(im, tuple(int(x * 255) for x in image_processor.image_mean))
            out.append(image_processor.preprocess(im, return_tensors="pt")["pixel_values"][0])
        if all(x.shape == out[0].shape for x in out):
            return torch.stack(out, 0)
        return out
    return image_processor(images, return_tensors="pt")["pixel_values"]
