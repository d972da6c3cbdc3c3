"""Exact / integer glue of the path: the helpers of util/util.py
and ppst_model.py that must be bit-identical to the reference given identical inputs."""
import torch

from . import ops


def normalize(v):
    """util.normalize (util/util.py:18-22) via ppst_l2norm_rows."""
    if isinstance(v, (list, tuple)):
        return [normalize(vv) for vv in v]
    return ops.l2norm_rows(v, 1e-8, 0)


def lerp(a, b, r):
    """util.lerp (util/util.py:32-35): a*(1-r) + b*r, lists element-wise."""
    if isinstance(a, (list, tuple)):
        return [lerp(aa, bb, r) for aa, bb in zip(a, b)]
    return ops.lerp(a, b, r)


def swap(x):
    """PPSTModel.swap (ppst_model.py:59-66): exchange the two members of every adjacent
    pair of the minibatch (pure index permutation, no arithmetic)."""
    shape = x.shape
    assert shape[0] % 2 == 0, "Minibatch size must be a multiple of 2"
    return torch.flip(x.reshape(shape[0] // 2, 2, *shape[1:]), [1]).reshape(shape)


def tensor2im(x):
    """util.tensor2im(tile=False) (util/util.py:98-131) for (B,C,H,W): uint8 HWC, truncation."""
    return ops.tensor2im_u8(x)


def one_hot_mask(labels, n=3):
    """CelebAMask_dataset.py:54-60: integer labels (B,H,W) in {0..n-1} -> float one-hot (B,n,H,W)."""
    return torch.stack([(labels == i) for i in range(n)], dim=1).float()


def gan_loss(pred, should_be_classified_as_real):
    """models/networks/loss.py:11-18 (LSGAN)."""
    return torch.mean((pred - 1) ** 2) if should_be_classified_as_real else torch.mean(pred ** 2)
