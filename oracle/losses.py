"""Oracle (test infrastructure): GAN objectives and the compositing formulas of
the ``geomgm_ifw_fore`` train step.

Restates Module2/models/networks.py:407-473 (GANLoss, lsgan),
Module2/models/base_model.py:238-247 (masked) and
Module2/models/geomgm_ifw_fore_model.py:523-543,589-635 (compositing, D losses).
"""
import torch


def gan_loss_lsgan(pred, target_is_real):
    """GANLoss('lsgan'): MSELoss(pred, 1.0 or 0.0 expanded) (networks.py:429-430,449-467)."""
    t = 1.0 if target_is_real else 0.0
    return ((pred - t) ** 2).mean()


def masked(a, mask, mask_type=3):
    """base_model.py:238-247."""
    if mask_type == 0:
        return (a / 2 + 0.5) * mask * 2 - 1
    if mask_type == 1:
        return ((a / 2 + 0.5) * mask + 1 - mask) * 2 - 1
    if mask_type == 2:
        return torch.cat((a, mask), 1)
    if mask_type == 3:
        m = ((a / 2 + 0.5) * mask + 1 - mask) * 2 - 1
        return torch.cat((m, mask), 1)
    raise ValueError(mask_type)


def fore_composite(x, mask):
    """Foreground on white: ((x/2+.5)*mask + 1-mask)*2-1 (geomgm_ifw_fore_model.py:523-527)."""
    return ((x / 2 + 0.5) * mask + 1 - mask) * 2 - 1


def bg_blend(fake, static, mask):
    """((f/2+.5)*m + (s/2+.5)*(1-m))*2-1 (geomgm_ifw_fore_model.py:541,543)."""
    return ((fake / 2 + 0.5) * mask + (static / 2 + 0.5) * (1 - mask)) * 2 - 1


def d_loss_basic(pred_real, pred_fake):
    """backward_D_basic, :567-587."""
    return (gan_loss_lsgan(pred_real, True) + gan_loss_lsgan(pred_fake, False)) * 0.5


def d_loss_basic2(pred_real, pred_fake1, pred_fake2):
    """backward_D_basic2, :589-611 (temporal-coherence D: real pair / fake pair / unrelated real pair)."""
    return (gan_loss_lsgan(pred_real, True) + gan_loss_lsgan(pred_fake1, False)
            + gan_loss_lsgan(pred_fake2, False)) / 3.0


def d_loss_basic3(pred_real, pred_fake1, pred_fake2):
    """backward_D_basic3, :613-635."""
    return (gan_loss_lsgan(pred_real, True)
            + (gan_loss_lsgan(pred_fake1, False) + gan_loss_lsgan(pred_fake2, False)) / 2.0) / 2.0
