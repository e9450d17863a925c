"""torchrun entry (reference finetuning/llava/train/train_mem.py:1-4)."""
from .train import train

if __name__ == "__main__":
    train()
