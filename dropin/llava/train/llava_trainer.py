from radvlm_amd.llava.train.llava_trainer import *  # noqa: F401,F403
