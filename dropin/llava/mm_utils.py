from radvlm_amd.llava.mm_utils import *  # noqa: F401,F403
