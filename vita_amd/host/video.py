"""Video front end of the offline demo: frame sampling and pre-processing of
`_get_rawvideo_dec` (video_audio_demo.py:30-118) — 1 frame per second (video_framerate), clamped to
[min_frames, max_frames] = [4, MAX_IMAGE_LENGTH = 16] by uniform re-sampling, optional [s, e] second window,
frames padded to square with the processor's mean colour ("pad") and pushed through the image processor.
Each frame then takes one IMAGE sentinel in the prompt (256 tokens after the tower).  The reader is anything
with decord's `get_avg_fps() / len() / get_batch(idx).asnumpy()`."""
import numpy as np
import torch
from PIL import Image

from .constants import MAX_IMAGE_LENGTH
from .image_processing import expand2square


def sample_positions(n_total, fps, max_frames=MAX_IMAGE_LENGTH, min_frames=4, video_framerate=1, s=None, e=None):
    """frame indices the reference decodes (video_audio_demo.py:43-77); [] if the window is empty."""
    if s is None:
        start_time = end_time = None
    else:
        start_time, end_time = int(s), int(e)
        start_time = start_time if start_time >= 0.0 else 0.0
        end_time = end_time if end_time >= 0.0 else 0.0
        if start_time > end_time:
            start_time, end_time = end_time, start_time
        elif start_time == end_time:
            end_time = start_time + 1
    f_start = 0 if start_time is None else int(start_time * fps)
    f_end = int(min(1000000000 if end_time is None else end_time * fps, n_total - 1))
    if f_end - f_start + 1 <= 0:
        return []
    t_stride = int(round(float(fps) / int(video_framerate)))
    all_pos = list(range(f_start, f_end + 1, t_stride))
    if len(all_pos) > max_frames:
        return [all_pos[i] for i in np.linspace(0, len(all_pos) - 1, num=max_frames, dtype=int)]
    if len(all_pos) < min_frames:
        return [all_pos[i] for i in np.linspace(0, len(all_pos) - 1, num=min_frames, dtype=int)]
    return all_pos


def get_rawvideo(reader, image_processor, max_frames=MAX_IMAGE_LENGTH, min_frames=4, video_framerate=1, s=None, e=None,
                 image_aspect_ratio="pad"):
    """-> (Tensor[n_frames, 3, H, W], n_frames), the pair `_get_rawvideo_dec` returns."""
    pos = sample_positions(len(reader), reader.get_avg_fps(), max_frames, min_frames, video_framerate, s, e)
    if not pos:
        raise ValueError("empty frame window")
    frames = [Image.fromarray(f) for f in reader.get_batch(pos).asnumpy()]
    if image_aspect_ratio == "pad":
        bg = tuple(int(x * 255) for x in image_processor.image_mean)
        frames = [expand2square(f, bg) for f in frames]
    out = torch.stack([image_processor.preprocess(f, return_tensors="pt")["pixel_values"][0] for f in frames])
    return out, out.shape[0]
