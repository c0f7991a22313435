"""Stand-in for the third-party `decord` package the reference demo imports at module level
(video_audio_demo.py:9) — it is not installed in this image and there is no video codec library here.

`VideoReader` implements the subset the reference uses (`get_avg_fps()`, `len()`, `get_batch(idx).asnumpy()`)
for containers that need no codec: a directory of frame images, an .npz/.npy frame array ([T,H,W,3] uint8,
optional `fps`), or any multi-frame image PIL can open (GIF / APNG / WebP / TIFF).  Compressed video
(mp4, mkv, ...) raises a clear error."""
import os

import numpy as np


def cpu(index=0):
    return ("cpu", index)


class _Batch:
    def __init__(self, arr):
        self._a = arr

    def asnumpy(self):
        return self._a


class VideoReader:
    IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp")

    def __init__(self, uri, ctx=None, **_):
        from PIL import Image
        self._frames, self._fps = None, 1.0
        if os.path.isdir(uri):
            files = sorted(f for f in os.listdir(uri) if f.lower().endswith(self.IMAGE_EXT))
            if not files:
                raise RuntimeError(f"{uri}: no frame images")
            self._frames = [np.asarray(Image.open(os.path.join(uri, f)).convert("RGB")) for f in files]
            fps_file = os.path.join(uri, "fps.txt")
            if os.path.exists(fps_file):
                self._fps = float(open(fps_file).read().strip())
        elif uri.lower().endswith(".npz"):
            z = np.load(uri)
            self._frames = list(np.asarray(z["frames"], np.uint8))
            self._fps = float(z["fps"]) if "fps" in z else 1.0
        elif uri.lower().endswith(".npy"):
            self._frames = list(np.asarray(np.load(uri), np.uint8))
        else:
            try:
                im = Image.open(uri)
                n = getattr(im, "n_frames", 1)
            except Exception as e:
                raise RuntimeError(f"{uri}: compressed video needs the real `decord` package (not installed here); "
                                   "supported without it: frame directory, .npz/.npy frame array, multi-frame image") from e
            frames = []
            for i in range(n):
                im.seek(i)
                frames.append(np.asarray(im.convert("RGB")))
            self._frames = frames
            dur = im.info.get("duration", 0)
            self._fps = 1000.0 / dur if dur else 1.0

    def __len__(self):
        return len(self._frames)

    def get_avg_fps(self):
        return self._fps

    def get_batch(self, indices):
        return _Batch(np.stack([self._frames[int(i)] for i in indices], 0))
