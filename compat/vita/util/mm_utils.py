"""vita/util/mm_utils.py of the reference."""
from vita_amd.host.image_processing import expand2square, process_images  # noqa: F401
from vita_amd.host.prompt import (KeywordsStoppingCriteria, get_model_name_from_path,  # noqa: F401
                                  tokenizer_image_audio_token, tokenizer_image_token)
