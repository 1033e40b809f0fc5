"""reference bird_view/utils/train_utils.py:33-40 (the training scripts import it as `train_util`)."""
import torch


def one_hot(x, num_digits=4, start=1):
    n = x.size()[0]
    x = torch.clamp(x.long()[:, None] - start, 0, num_digits - 1)
    y = torch.zeros(n, num_digits, dtype=torch.float32)
    y.scatter_(1, x, 1)
    return y
