from radvlm_amd.llava.train.train import *  # noqa: F401,F403
from radvlm_amd.llava.train.train import train  # noqa: F401

if __name__ == "__main__":
    train()
