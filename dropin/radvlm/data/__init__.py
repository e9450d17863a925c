from radvlm_amd.data import *  # noqa: F401,F403
