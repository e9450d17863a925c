from radvlm_amd.llava.conversation import *  # noqa: F401,F403
from radvlm_amd.llava.conversation import conv_templates, default_conversation  # noqa: F401
