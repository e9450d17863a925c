"""Drop-in alias: put this directory on PYTHONPATH and `import llava...` resolves to the MI355X build."""
from radvlm_amd.llava import *  # noqa: F401,F403
from radvlm_amd.llava import constants, conversation, mm_utils  # noqa: F401
