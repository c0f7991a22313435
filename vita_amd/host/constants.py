"""Sentinels and limits shared by the host-side prompt code (values from vita/constants.py:2-9)."""
MAX_IMAGE_LENGTH = 16
MIN_IMAGE_LENGTH = 4
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
AUDIO_TOKEN_INDEX = -500
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_VIDEO_TOKEN = "<video>"
DEFAULT_AUDIO_TOKEN = "<audio>"
GLOBAL_WEIGHTS_PATH = ""
