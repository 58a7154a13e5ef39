#!/usr/bin/env python3
"""Golden vectors for the glue around the frozen auxiliary networks (tests/golden/aux.npz), produced by running the
REFERENCE's own functions (imported read-only from /root/reference/Module2) in the build container:

* ``GeomGMIFWForeModel.get_lm``       geomgm_ifw_fore_model.py:390-415   (crop + bicubic + landmark net + re-projection)
* ``networks.FaceLoss.forward``       networks.py:2881-2966              (crop + bilinear 112x96 + feature L1)
* ``kp_to_map_some``                  geomgm_ifw_fore_model.py:19-51
* ``flow_network_warp``               geomgm_ifw_fore_model.py:69-84

    python tests/golden/make_aux_golden.py

The frozen nets' checkpoints are not in the reference tree, so the fixed-seed stand-ins of
``animateportrait_amd/standins.py`` (same call contracts) take their place on both sides.  Import shims: an empty
``cv2`` module (only ``getlipline`` / ``get_lmvis`` use it), the ``torchvision`` / ``skimage`` stubs of make_golden.py,
and ``Tensor.cuda`` as the identity (the reference hard-codes ``.cuda()``; this container has no GPU).  Methods are
called unbound on a namespace that carries the attributes they read.  Nothing of the reference's source is stored.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    import warnings
    warnings.filterwarnings('ignore')
    from make_golden import import_reference, save
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    networks, _, _, _ = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models import geomgm_ifw_fore_model as ref
    from animateportrait_amd import standins
    torch.set_num_threads(8)
    out = {}
    g = torch.Generator().manual_seed(4321)

    # ---- get_lm: b=1 (the reference's assumption), 1-channel drawing and 3-channel photo, windows partly outside
    class Rec(torch.nn.Module):
        """records what the landmark net is fed"""
        def __init__(self, net):
            super().__init__()
            self.net, self.seen = net, []

        def forward(self, x):
            self.seen.append(x.detach().clone())
            return self.net(x)
    lmnet = Rec(standins.StandinLandmarkNet())
    fake_self = types.SimpleNamespace(opt=types.SimpleNamespace(crop_size=256), mobilefacenet=lmnet, gpu=0, gpu_p=0)
    wins = [[32, 224, 32, 224], [-20, 200, 10, 230], [90, 290, 60, 250]]
    for c in (1, 3):
        for i, w in enumerate(wins):
            seed = 9000 + 10 * c + i                                     # inputs are regenerated from the seed
            x = torch.rand(1, c, 256, 256, generator=torch.Generator().manual_seed(seed)) * 2 - 1
            win = torch.IntTensor([w])
            lm = ref.GeomGMIFWForeModel.get_lm(fake_self, x, win)
            out['lm_c%d_%d_seed' % (c, i)] = np.int64(seed)
            out['lm_c%d_%d_win' % (c, i)] = win.numpy()
            box = lmnet.seen[-1]
            # c=1: the three box channels are copies; c=3: keep every channel for the window that leaves the image
            out['lm_c%d_%d_box' % (c, i)] = box[:, :1] if (c == 1 or i != 1) else box
            out['lm_c%d_%d_boxsum' % (c, i)] = box.double().sum(dim=(0, 2, 3))
            out['lm_c%d_%d_out' % (c, i)] = lm.detach()

    # ---- FaceLoss: real class, constructor bypassed (it only loads the absent checkpoint), net = stand-in
    fnet = Rec(standins.StandinFaceNet())
    fl = networks.FaceLoss.__new__(networks.FaceLoss)
    torch.nn.Module.__init__(fl)
    fl.net, fl.height, fl.width, fl.criterion = fnet, 112, 96, torch.nn.L1Loss()
    a = torch.rand(2, 1, 256, 256, generator=torch.Generator().manual_seed(9100)) * 2 - 1
    b = torch.rand(2, 1, 256, 256, generator=torch.Generator().manual_seed(9101)) * 2 - 1
    bb1 = torch.IntTensor([[32, 224, 32, 224], [-10, 180, 20, 200]])
    bb2 = torch.IntTensor([[40, 230, 30, 210], [60, 270, 70, 256]])
    a3 = a.repeat(1, 3, 1, 1).requires_grad_(True)
    loss = fl(a3, b.repeat(1, 3, 1, 1), bbox1=bb1, bbox2=bb2)
    loss.backward()
    ga = a3.grad.sum(1, keepdim=True)                                   # d loss / d (the 1-channel drawing)
    out.update(fl_seed_a=np.int64(9100), fl_seed_b=np.int64(9101), fl_bb1=bb1.numpy(), fl_bb2=bb2.numpy(),
               fl_head1=fnet.seen[0][:, :1], fl_head2=fnet.seen[1][:, :1], fl_loss=loss.detach(),
               fl_grad_a_sub=ga[:, :, ::4, ::4], fl_grad_a_sum=ga.double().sum(), fl_grad_a_abs=ga.double().abs().sum())

    # ---- kp_to_map_some and flow_network_warp (stand-in FlowUnet)
    lm1 = torch.rand(1, 68, 2, generator=g) * 200 + 28
    lm2 = lm1 + torch.randn(1, 68, 2, generator=g) * 3
    lm1[0, 5] = torch.tensor([255.0, 0.0])                       # corner cases: on the border / exactly integer
    lm1[0, 6] = torch.tensor([128.0, 64.0])
    j1 = ref.kp_to_map_some((224, 224), lm1.numpy() * 7 / 8)
    out.update(kp_lm1=lm1, kp_lm2=lm2, kp_j1=j1.numpy().astype(np.uint8))
    netF = standins.StandinFlowNet()
    real_A = torch.zeros(1, 3, 256, 256)
    wf, rm = ref.flow_network_warp(netF, real_A, lm1, lm2)
    out.update(fw_flow_sub=wf[:, :, ::2, ::2], fw_mask_sub=rm[:, :, ::2, ::2], fw_flow_abs=wf.double().abs().sum(),
               fw_mask_sum=rm.double().sum())
    save('aux.npz', **out)


if __name__ == '__main__':
    main()
