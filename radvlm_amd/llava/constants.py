"""Names of the LLaVA data/model contract that importers of ``llava.constants`` expect (reference finetuning/llava/constants.py:
token strings inside LLaVA-format JSON, the label / placeholder ids the collator and the splice agree on)."""
from radvlm_amd.splice import IGNORE_INDEX, IMAGE_TOKEN_INDEX  # -100: label ignored by the loss; -200: image placeholder id

DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"

# serving-side knobs some reference modules import from here; unused on the training path
LOGDIR = "."
WORKER_HEART_BEAT_INTERVAL = 15
CONTROLLER_HEART_BEAT_EXPIRATION = 30

__all__ = ["IGNORE_INDEX", "IMAGE_TOKEN_INDEX", "DEFAULT_IMAGE_TOKEN", "DEFAULT_IMAGE_PATCH_TOKEN", "DEFAULT_IM_START_TOKEN",
           "DEFAULT_IM_END_TOKEN", "LOGDIR", "WORKER_HEART_BEAT_INTERVAL", "CONTROLLER_HEART_BEAT_EXPIRATION"]
