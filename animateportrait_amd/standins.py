"""Stand-ins for the frozen third-party networks of the train step.

The reference's ``backward_G`` / ``set_input`` call four pretrained nets whose checkpoints are NOT in its tree
(SURVEY.md section 2 row 12): MobileFaceNet (68-point landmark regressor, geomgm_ifw_fore_model.py:362-368,
called at :410), Sphere20a inside ``FaceLoss`` (identity features, networks.py:2862-2878), MODNet (matte,
:369-373) and FlowUnet ``netF`` (intrinsic flow, :57-68).  On the product path they are ``model.aux`` callables
-- stock PyTorch-ROCm modules the user loads.  This file holds small fixed-seed networks with the SAME call
contracts, so that the code around them (window crop + bicubic resize, crop + bilinear resize to 112x96, the
five-feature L1, ``kp_to_map`` / flow post-processing, the gradient flowing back into the generator) can be
executed, tested against the oracle / the reference's own functions, and timed.
They are NOT models of anything: weights are N(0, s) under a private seed.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _seeded(module, seed, std):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (std if p.dim() > 1 else 0.01))
            p.requires_grad_(False)            # frozen, like the pretrained nets they stand for
    return module.eval()


class StandinLandmarkNet(nn.Module):
    """Contract of MobileFaceNet([112, 112], 136) as ``get_lm`` uses it (geomgm_ifw_fore_model.py:410):
    ``net(x)[0]`` with x (B, 3, 112, 112) in [0, 1] -> (B, 136) window-normalised (x, y) pairs."""

    def __init__(self, seed=101):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, 2, 1)
        self.c2 = nn.Conv2d(8, 16, 3, 2, 1)
        self.fc = nn.Linear(16 * 4 * 4, 136)
        _seeded(self, seed, 0.15)

    def forward(self, x):
        f = F.relu(self.c2(F.relu(self.c1(x))))
        out = torch.sigmoid(self.fc(F.adaptive_avg_pool2d(f, 4).flatten(1)))
        # a copy: the reference's get_lm re-projects the regressor's output IN PLACE (:412-413), which autograd only allows
        # when the producing op does not keep its output for backward (MobileFaceNet ends in a Linear; sigmoid keeps it)
        return out.clone(), f


class StandinFaceNet(nn.Module):
    """Contract of Sphere20a (models/facenet.py:200-282) inside FaceLoss (networks.py:2926-2940):
    x (B, 3, 112, 96) in [-1, 1] -> list of five feature tensors (four maps at /2 /4 /8 /16, one vector)."""

    def __init__(self, seed=102):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, 2, 1)
        self.c2 = nn.Conv2d(8, 12, 3, 2, 1)
        self.c3 = nn.Conv2d(12, 16, 3, 2, 1)
        self.c4 = nn.Conv2d(16, 16, 3, 2, 1)
        self.fc = nn.Linear(16 * 7 * 6, 32)
        self.a = nn.PReLU(1)
        _seeded(self, seed, 0.2)
        with torch.no_grad():
            self.a.weight.fill_(0.25)

    def forward(self, x):
        feats = []
        for conv in (self.c1, self.c2, self.c3, self.c4):
            x = self.a(conv(x))
            feats.append(x)
        feats.append(self.fc(x.flatten(1)))
        return feats


class StandinFlowNet(nn.Module):
    """Contract of FlowUnet_v2 as ``flow_network_warp`` calls it (geomgm_ifw_fore_model.py:69-84):
    (B, 136, 224, 224) binary joint maps -> (flow_out (B,2,224,224), vis_out (B,3,224,224) logits, None, None)."""

    def __init__(self, seed=103):
        super().__init__()
        self.c1 = nn.Conv2d(136, 8, 5, 1, 2)
        self.flow = nn.Conv2d(8, 2, 3, 1, 1)
        self.vis = nn.Conv2d(8, 3, 3, 1, 1)
        _seeded(self, seed, 0.3)

    def forward(self, x):
        f = torch.tanh(F.avg_pool2d(self.c1(x), 9, 1, 4))
        return self.flow(f), self.vis(f), None, None


class StandinMatteNet(nn.Module):
    """Contract of MODNet as the models call it: ``modnet(img, True)`` -> (_, _, matte (B,1,H,W) in [0,1])
    (geomcgt_ifw_test_model.py:279) / ``modnet(img)`` used through ``> 0.5`` (geomgm_ifw_fore_model.py:519-521)."""

    def __init__(self, seed=104):
        super().__init__()
        self.c = nn.Conv2d(3, 1, 9, 1, 4)
        _seeded(self, seed, 0.5)

    def forward(self, x, inference=True):
        h, w = x.shape[2:]
        yy = torch.linspace(-1, 1, h, device=x.device).view(1, 1, h, 1)
        xx = torch.linspace(-1, 1, w, device=x.device).view(1, 1, 1, w)
        matte = torch.sigmoid(4.0 - 8.0 * (xx * xx + yy * yy) + 0.5 * self.c(x))     # a soft disc, perturbed
        return None, None, matte
