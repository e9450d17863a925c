from radvlm_amd.llava.constants import *  # noqa: F401,F403
