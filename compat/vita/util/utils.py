"""vita/util/utils.py of the reference: disable_torch_init (vita/util/utils.py:14-21) skips the default
nn.Linear / nn.LayerNorm initialisers to speed model construction; the HIP model owns no nn.Parameter, so
there is nothing to skip — kept as a callable for the demo (video_audio_demo.py:160)."""


def disable_torch_init():
    return None
