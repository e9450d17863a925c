from radvlm_amd.llava.train.train import train

if __name__ == "__main__":
    train()
