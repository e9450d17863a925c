"""Names of the LLaVA data/model contract that importers of ``llava.constants`` expect.

The values are interface facts (token strings inside LLaVA-format JSON, the label / placeholder ids the collator and the
splice agree on; reference finetuning/llava/constants.py), restated here from one table."""
from radvlm_amd.splice import IGNORE_INDEX, IMAGE_TOKEN_INDEX  # -100: label ignored by the loss; -200: image placeholder id

_TOKENS = {"IMAGE": "<image>", "IMAGE_PATCH": "<im_patch>", "IM_START": "<im_start>", "IM_END": "<im_end>"}
(DEFAULT_IMAGE_TOKEN, DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IM_END_TOKEN) = (
    _TOKENS["IMAGE"], _TOKENS["IMAGE_PATCH"], _TOKENS["IM_START"], _TOKENS["IM_END"])

# serving-side knobs some reference modules import from here; unused on the training path
LOGDIR, WORKER_HEART_BEAT_INTERVAL, CONTROLLER_HEART_BEAT_EXPIRATION = ".", 15, 30

__all__ = ["IGNORE_INDEX", "IMAGE_TOKEN_INDEX", "DEFAULT_IMAGE_TOKEN", "DEFAULT_IMAGE_PATCH_TOKEN", "DEFAULT_IM_START_TOKEN",
           "DEFAULT_IM_END_TOKEN", "LOGDIR", "WORKER_HEART_BEAT_INTERVAL", "CONTROLLER_HEART_BEAT_EXPIRATION"]
