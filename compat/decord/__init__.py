"""Stand-in for the third-party `decord` package the reference demo imports at module level
(video_audio_demo.py:9).  It is not installed in this image; only --video_path needs it."""


def cpu(index=0):
    return ("cpu", index)


class VideoReader:
    def __init__(self, *args, **kwargs):
        raise ImportError("decord is not installed in this environment: video input (--video_path) is unavailable; "
                          "image / audio / text prompts do not need it")
