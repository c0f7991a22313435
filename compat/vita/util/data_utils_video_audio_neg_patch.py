"""vita/util/data_utils_video_audio_neg_patch.py:1197-1255 of the reference (inference-side subset)."""
from vita_amd.host.image_processing import dynamic_preprocess, find_closest_aspect_ratio  # noqa: F401
