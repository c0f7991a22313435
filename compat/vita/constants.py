"""vita/constants.py of the reference."""
from vita_amd.host.constants import *  # noqa: F401,F403

CONTROLLER_HEART_BEAT_EXPIRATION = 30
LOGDIR = "gradio-logs"
WORKER_HEART_BEAT_INTERVAL = 15
DEFAULT_DATA_RATIO = 1.0
