"""``--model geomgm_ifw_fore`` on the MI355X: the training model of
Module2/models/geomgm_ifw_fore_model.py (options :161-209, __init__ :211-388, set_input :443-505,
forward :517-565, backward_D_* :567-672, backward_G :677-780, optimize_parameters :782-819).

What is the same: option names/defaults, network construction through ``networks.define_G/define_D``,
``model_names`` / ``loss_names`` / checkpoint naming, every loss formula and weight, the order
G-step then D-step, Adam(lr, (beta1, 0.999)) for G and for the concatenation of the D parameters.

What is different (and why):
* Batched.  The reference train step silently assumes batch size 1 (``[0]`` indexing at :391-398, :509; a
  ``bmm`` that only broadcasts for b=1 in sparse_image_warp.py:201-203).  Here every formula is applied per
  sample and mean-reduced losses average over the batch -- exact for InstanceNorm networks.
* The two generator calls of :530-531 (and the paired discriminator calls) run as ONE launch over a 2B batch;
  InstanceNorm makes this identical to two calls.
* One process per GPU; gradients are averaged by ``parallel.allreduce_optimizer_grads`` (RCCL) where the
  reference relies on nn.DataParallel.  ``--gpu_ids_p`` is accepted and ignored (everything lives on the
  rank's GPU).
* The frozen auxiliary networks (MODNet matte, MobileFaceNet landmarks, Sphere20a identity features, FlowUnet
  intrinsic flow; :57-84, :362-377) are third-party nets whose checkpoints are not in the reference tree
  (SURVEY.md section 2, row 12).  They enter through ``self.aux`` callables (stock PyTorch-ROCm modules the user
  loads); when one is absent its OUTPUT is taken from the batch dict (``mask``, ``iw_flow``, ``if_mask`` ...) and
  the loss terms that need it (geometry, identity) are skipped with a one-time notice.
"""
import itertools

import torch

from .. import losses, networks, parallel
from ..autograd import direct_param_grads
from ..optim import FlatAdam
from ..util.image_pool import ImagePool
from .base_model import BaseModel
from .sparse_image_warp import warp_nchw

# the first 20 segments of the reference's faceLmarkLookup.npy: outer (48..59) and inner (60..67) lip loops
LIP_SEGMENTS = [(i, i + 1) for i in range(48, 59)] + [(59, 48)] + [(i, i + 1) for i in range(60, 67)] + [(67, 60)]


class GeomGMIFWForeModel(BaseModel):
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        parser.set_defaults(no_dropout=True)
        parser.set_defaults(dataset_mode='umlvd_ifw')
        parser.set_defaults(netG='resnet_9blocks_rcatland3')
        parser.add_argument('--netg_resb_div', type=int, default=3)
        parser.add_argument('--netg_resb_disp', type=int, default=1)
        if is_train:
            parser.add_argument('--lambda_geom', type=float, default=5.0)
            parser.add_argument('--lambda_geom_lipline', type=float, default=0.0)
            parser.add_argument('--max_offset', type=float, default=3)
            parser.add_argument('--lambda_G_A_l', type=float, default=0.5)
        parser.add_argument('--use_mask', type=int, default=1)
        parser.add_argument('--use_eye_mask', type=int, default=1)
        parser.add_argument('--use_lip_mask', type=int, default=1)
        parser.add_argument('--mask_type', type=int, default=3)
        parser.add_argument('--blendbg', type=int, default=0)
        if is_train:
            parser.add_argument('--identity_loss', type=int, default=2)
            parser.add_argument('--face_recog_model', type=str, default='./checkpoints/sphere20a_20171020.pth')
            parser.add_argument('--lambda_face', type=float, default=5.0)
            parser.add_argument('--warp_loss', type=int, default=2)
            parser.add_argument('--lambda_warp', type=float, default=5.0)
            parser.add_argument('--lambda_warp_inter', type=float, default=5.0)
            parser.add_argument('--coherent', type=int, default=1)
            parser.add_argument('--lambda_G_A_coh', type=float, default=0.5)
            parser.add_argument('--coh_use_more', type=int, default=2)
            parser.add_argument('--check_fakeb2_in_backwardD', type=int, default=1)
            parser.add_argument('--select_target12_thre', type=float, default=0.2)
            parser.add_argument('--select_noniden_thre', type=float, default=0.9)
            parser.add_argument('--more_weight_for_lip', type=int, default=0)
            parser.add_argument('--rx', type=float, default=0.15)
            parser.add_argument('--ry', type=float, default=0.2)
            parser.add_argument('--rs', type=float, default=0.7)
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.loss_names = ['D_A', 'G_A', 'G']                                    # :217-247
        if self.isTrain and opt.use_mask:
            self.loss_names += ['D_A_l', 'G_A_l']
        if self.isTrain and opt.use_eye_mask:
            self.loss_names += ['D_A_le', 'G_A_le']
        if self.isTrain and opt.use_lip_mask:
            self.loss_names += ['D_A_ll', 'G_A_ll']
        if self.isTrain and opt.coherent:
            self.loss_names += ['D_A_coh', 'G_A_coh']
        if self.isTrain:
            self.loss_names += ['geom_B', 'geom_B_lipline', 'warp_B', 'warp_inter1', 'iden_B']
        self.visual_names = ['real_A', 'fake_B', 'fake_B2', 'real_B']
        self.model_names = ['G_A']
        if self.isTrain:                                                         # :288-299
            self.model_names += ['D_A']
            self.model_names += ['D_A_l'] if opt.use_mask else []
            self.model_names += ['D_A_le'] if opt.use_eye_mask else []
            self.model_names += ['D_A_ll'] if opt.use_lip_mask else []
            self.model_names += ['D_A_coh'] if opt.coherent else []
        gid = [self.gpu_ids[0]]
        self.netG_A = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm, not opt.no_dropout,
                                        opt.init_type, opt.init_gain, gid, div=opt.netg_resb_div,
                                        disp=opt.netg_resb_disp)                 # :301-302
        self.aux = {'modnet': None, 'landmarks': None, 'faceloss': None, 'netF': None}
        self._noticed = set()
        if self.isTrain:
            def D(nc):
                return networks.define_D(nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm, opt.init_type,
                                         opt.init_gain, gid)
            local_nc = opt.output_nc + 1 if opt.mask_type in (2, 3) else opt.output_nc
            self.netD_A = D(opt.output_nc)                                       # :312-313
            if opt.use_mask:
                self.netD_A_l = D(local_nc)
            if opt.use_eye_mask:
                self.netD_A_le = D(local_nc)
            if opt.use_lip_mask:
                self.netD_A_ll = D(local_nc)
            if opt.coherent:
                self.netD_A_coh = D(opt.output_nc * 2)                           # :336
            self.fake_B_pool = ImagePool(opt.pool_size)
            self.criterionGAN = networks.GANLoss(opt.gan_mode).to(self.device)
            self.criterionIdt = torch.nn.L1Loss()
            d_params = list(itertools.chain(*[getattr(self, 'net' + n).parameters() for n in self.model_names[1:]]))
            self.optimizer_G = FlatAdam(self.netG_A.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))   # :346
            self.optimizer_D = FlatAdam(d_params, lr=opt.lr, betas=(opt.beta1, 0.999))                   # :347-360
            self.optimizers += [self.optimizer_G, self.optimizer_D]
        cs = opt.crop_size
        csh = cs // 2
        self.edges = torch.tensor([[[0, 0], [cs - 1, cs - 1], [0, cs - 1], [cs - 1, 0], [0, csh - 1], [csh - 1, 0],
                                    [csh - 1, cs - 1], [cs - 1, csh - 1]]], dtype=torch.float32, device=self.device)
        self.thickness = 4 if cs == 512 else 2                                   # :381-385

    # ------------------------------------------------------------------ helpers
    def _notice(self, key, msg):
        if key not in self._noticed:
            self._noticed.add(key)
            print('[geomgm_ifw_fore] ' + msg)

    def update_process(self, epoch):                                             # :674-675
        self.process = (epoch - 1) / float(self.opt.niter_decay + self.opt.niter)

    def getlipline(self, lands):
        """:507-515: the 20 lip segments drawn with cv2.line(thickness 2 at 256 px, 4 at 512 px, value 1), per sample --
        OpenCV's ThickLine rule run on the device (ap_lip_line_mask; restated with its source in oracle/cv_raster.py)."""
        return losses.lip_line_mask(lands.to(self.device), LIP_SEGMENTS, self.opt.crop_size, self.thickness)

    def get_lm(self, x, win, out_size=112):
        """:390-415, per sample: window crop into a ones-filled box, BGR / x3 channels, bicubic (align_corners=False)
        to 112^2, [-1,1] -> [0,1] -- one fused HIP launch each way (ap_crop_resize_*) -- then the frozen landmark
        regressor and the re-projection of its window-normalised output to image pixels."""
        net = self.aux['landmarks']
        n, c = x.shape[:2]
        wdev = losses.windows_to_device(win, n, x.device)
        box = losses.crop_resize(x, wdev, (2, 1, 0) if c == 3 else (0, 0, 0), (out_size, out_size),
                                 losses.RESIZE_BICUBIC, scale=0.5, shift=0.5)
        lm = net(box)
        lm = (lm[0] if isinstance(lm, (tuple, list)) else lm).view(n, 68, 2)
        wf = wdev.to(x.dtype)
        scale = torch.stack([wf[:, 1] - wf[:, 0], wf[:, 3] - wf[:, 2]], 1).view(n, 1, 2)
        off = torch.stack([wf[:, 0], wf[:, 2]], 1).view(n, 1, 2)
        return lm * scale + off

    # ------------------------------------------------------------------ data
    def set_input(self, input):
        """:443-505.  Same dict keys as UMLVDIFWDataset.__getitem__; tensors may already live on the GPU."""
        def dev(k):
            return input[k].to(self.device, non_blocking=True).float().contiguous()
        AtoB = self.opt.direction == 'AtoB'
        self.real_A = dev('A' if AtoB else 'B')
        self.real_B = dev('B' if AtoB else 'A')
        self.warp_motion, self.warp_motion2 = dev('warp_motion'), dev('warp_motion2')
        self.real_A_lm, self.target_B_lm, self.target_B2_lm = dev('A_lm'), dev('tB_lm'), dev('tB2_lm')
        self.real_A_lm_68, self.target_B_lm_68, self.target_B2_lm_68 = dev('A_lm_68'), dev('tB_lm_68'), dev('tB2_lm_68')
        self.winA, self.winB, self.winB2, self.winBr = (input[k] for k in ('winA', 'winB', 'winB2', 'winBr'))
        self.image_paths = input.get('image_paths', [])
        if self.isTrain and (self.opt.warp_loss == 2 or self.opt.identity_loss == 2):
            self.fakeB_static = dev('fakeB_static')
        elif self.isTrain and self.opt.warp_loss == 1:
            self.fakeB_static_warp = dev('fakeB_static_warp')
        for flag, suf in ((self.opt.use_mask, ''), (self.opt.use_eye_mask, 'e'), (self.opt.use_lip_mask, 'l')):
            if flag:
                for pre in ('Br', 'B', 'B2'):
                    setattr(self, '%s_mask%s' % (pre, suf), dev('%s_mask%s' % (pre, suf)))
        if self.isTrain and self.opt.coherent:
            self.real_B1, self.real_B2 = dev('B1'), dev('B2')
            if self.opt.coh_use_more:
                self.real_B3, self.real_B4 = dev('B3'), dev('B4')
        # step boundary: graphed aux nets forget forwards of the previous step whose backward never ran (aux_nets.GraphedFrozen)
        for a in self.aux.values():
            for m in ((a.modules() if isinstance(a, torch.nn.Module) else ()) if a is not None else ()):
                if hasattr(m, 'new_step'):
                    m.new_step()
        if self.aux['netF'] is not None:                                         # :503-505
            nf = self.aux['netF']                 # the frozen FlowUnet module itself; pre / post stages run on the device
            lm_a = self.real_A_lm_68[:, :68]
            b = self.real_A.shape[0]
            if getattr(nf, 'training', False):
                # a net in train mode (BatchNorm batch statistics, dropout) is not sample-independent: two B-sized calls, as the
                # reference makes them (the reference itself calls netF.eval(), :387; attach_flow_network does too -- ADVICE r5)
                self.iw_flow, self.real_A_if_mask = losses.flow_network_warp(nf, self.real_A, lm_a, self.target_B_lm_68[:, :68])
                self.iw_flow2, self.real_A_if_mask2 = losses.flow_network_warp(nf, self.real_A, lm_a, self.target_B2_lm_68[:, :68])
            else:
                # the reference calls flow_network_warp twice (photo -> target, photo -> second target); the frozen net is
                # sample-independent (eval mode), so both calls run as ONE 2B batch: half the launches on its small maps
                flow2, mask2 = losses.flow_network_warp(nf, self.real_A, torch.cat([lm_a, lm_a], 0),
                                                        torch.cat([self.target_B_lm_68[:, :68], self.target_B2_lm_68[:, :68]], 0))
                self.iw_flow, self.iw_flow2 = flow2[:b].contiguous(), flow2[b:].contiguous()
                self.real_A_if_mask, self.real_A_if_mask2 = mask2[:b].contiguous(), mask2[b:].contiguous()
        else:
            self._notice('netF', 'no intrinsic-flow network: iw_flow / if_mask are read from the batch')
            self.iw_flow, self.real_A_if_mask = dev('iw_flow'), dev('if_mask')
            self.iw_flow2, self.real_A_if_mask2 = dev('iw_flow2'), dev('if_mask2')
        if self.aux['modnet'] is not None:
            with torch.no_grad():
                _, _, matte = self.aux['modnet'](self.real_A, True)               # :519-521
                self.mask = (matte > 0.5).float()
        else:
            self._notice('modnet', 'no matting network: the foreground mask is read from the batch')
            self.mask = (dev('mask') > 0.5).float()

    # ------------------------------------------------------------------ forward
    def forward(self):
        """:517-565."""
        o = self.opt
        mask = self.mask
        fore = lambda x: losses.fore_composite(x, mask)                           # noqa: E731  (:523-527)
        if not o.blendbg:
            self.real_A = fore(self.real_A)
            self.real_A_fore = self.real_A
            if hasattr(self, 'fakeB_static'):
                self.fakeB_static = fore(self.fakeB_static)
        else:
            self.real_A_fore = fore(self.real_A)
        b = self.real_A.shape[0]
        cat2 = lambda a, c: torch.cat([a, c], 0)                                  # noqa: E731
        both = self.netG_A(cat2(self.real_A_fore, self.real_A_fore), cat2(self.real_A_lm, self.real_A_lm),
                           cat2(self.target_B_lm, self.target_B2_lm), cat2(self.warp_motion, self.warp_motion2),
                           cat2(self.iw_flow, self.iw_flow2), cat2(self.real_A_if_mask, self.real_A_if_mask2))
        self.fake_B, self.fake_B2 = both[:b], both[b:]                            # :530-531
        rc = lambda lm: lm[:, :, [1, 0]]                                          # noqa: E731  (x,y) -> (row,col)
        if o.blendbg:                                                             # :533-543
            self.real_A_lm_681, self.target_B_lm_681 = self.real_A_lm_68, self.target_B_lm_68
            m12 = warp_nchw(cat2(mask, mask), rc(cat2(self.real_A_lm_68, self.real_A_lm_68)),
                            rc(cat2(self.target_B_lm_68, self.target_B2_lm_68)))
            self.mask1, self.mask2 = m12[:b], m12[b:]
            self.fake_B_fore, self.fake_B2_fore = self.fake_B, self.fake_B2
            self.fake_B = losses.bg_blend(self.fake_B, self.fakeB_static, self.mask1)     # :541
            self.fake_B2 = losses.bg_blend(self.fake_B2, self.fakeB_static, self.mask2)   # :543
        for flag, suf in ((o.use_mask, ''), (o.use_eye_mask, 'e'), (o.use_lip_mask, 'l')):   # :546-557
            if flag:
                setattr(self, 'fake_B_l' + suf, self.masked(self.fake_B, getattr(self, 'B_mask' + suf)))
                setattr(self, 'fake_B2_l' + suf, self.masked(self.fake_B2, getattr(self, 'B2_mask' + suf)))
                setattr(self, 'real_B_l' + suf, self.masked(self.real_B, getattr(self, 'Br_mask' + suf)))
        if self.isTrain and o.warp_loss == 2:                                     # :558-565
            if not o.blendbg:       # edge points are only appended when blendbg did not already set *_681
                e = self.edges.expand(b, -1, -1)
                self.real_A_lm_681 = torch.cat((self.real_A_lm_68, e), 1)
                self.target_B_lm_681 = torch.cat((self.target_B_lm_68, e), 1)
            self.fakeB_static_warp = warp_nchw(self.fakeB_static, rc(self.real_A_lm_681), rc(self.target_B_lm_681))

    # ------------------------------------------------------------------ D losses
    def _d_loss3(self, netD, real, fake1, fake2):
        """backward_D_basic3 (:613-635) / backward_D_basic (:567-587); the fakes share one 2B launch."""
        if not self.opt.check_fakeb2_in_backwardD:
            loss = (self.criterionGAN(netD(real), True) + self.criterionGAN(netD(fake1.detach()), False)) * 0.5
        else:
            b = real.shape[0]
            pf = netD(torch.cat([fake1.detach(), fake2.detach()], 0))
            loss = (self.criterionGAN(netD(real), True)
                    + (self.criterionGAN(pf[:b], False) + self.criterionGAN(pf[b:], False)) / 2.0) / 2.0
        with direct_param_grads():          # a plain .backward() of the train step: accumulate straight into the flat buffers
            loss.backward()
        return loss.detach()

    def backward_D_A(self):
        self.loss_D_A = self._d_loss3(self.netD_A, self.real_B, self.fake_B, self.fake_B2)

    def backward_D_A_l(self):
        self.loss_D_A_l = self._d_loss3(self.netD_A_l, self.real_B_l, self.fake_B_l, self.fake_B2_l)

    def backward_D_A_le(self):
        self.loss_D_A_le = self._d_loss3(self.netD_A_le, self.real_B_le, self.fake_B_le, self.fake_B2_le)

    def backward_D_A_ll(self):
        self.loss_D_A_ll = self._d_loss3(self.netD_A_ll, self.real_B_ll, self.fake_B_ll, self.fake_B2_ll)

    def backward_D_A_coh(self):
        """:665-672 with backward_D_basic2 (:589-611): real pair / pooled fake pair / unrelated real pair."""
        fake_B = self.fake_B_pool.query(self.fake_B)
        fake_B2 = self.fake_B_pool.query(self.fake_B2)
        real = torch.cat((self.real_B1, self.real_B2), 1)
        fake = torch.cat((fake_B, fake_B2), 1).detach()
        D, crit = self.netD_A_coh, self.criterionGAN
        if not self.opt.coh_use_more:
            loss = (crit(D(real), True) + crit(D(fake), False)) * 0.5
        else:
            other = torch.cat((self.real_B3, self.real_B4), 1)
            b = real.shape[0]
            pf = D(torch.cat([fake, other], 0))
            loss = (crit(D(real), True) + crit(pf[:b], False) + crit(pf[b:], False)) / 3.0
        with direct_param_grads():          # a plain .backward() of the train step: accumulate straight into the flat buffers
            loss.backward()
        self.loss_D_A_coh = loss.detach()

    # ------------------------------------------------------------------ G loss
    def backward_G(self):
        """:677-780."""
        o, crit = self.opt, self.criterionGAN
        b = self.fake_B.shape[0]

        def gan2(netD, f1, f2):           # D(f1) and D(f2) as one 2B launch; sum of the two lsgan terms
            p = netD(torch.cat([f1, f2], 0))
            return crit(p[:b], True) + crit(p[b:], True)
        self.loss_G_A = gan2(self.netD_A, self.fake_B, self.fake_B2)
        loss = self.loss_G_A
        if o.use_mask:
            self.loss_G_A_l = gan2(self.netD_A_l, self.fake_B_l, self.fake_B2_l) * o.lambda_G_A_l
            loss = loss + self.loss_G_A_l
        if o.use_eye_mask:
            self.loss_G_A_le = gan2(self.netD_A_le, self.fake_B_le, self.fake_B2_le) * o.lambda_G_A_l
            loss = loss + self.loss_G_A_le
        if o.use_lip_mask:
            self.loss_G_A_ll = gan2(self.netD_A_ll, self.fake_B_ll, self.fake_B2_ll) * o.lambda_G_A_l
            loss = loss + self.loss_G_A_ll
        if o.coherent:
            self.loss_G_A_coh = crit(self.netD_A_coh(torch.cat((self.fake_B, self.fake_B2), 1)), True) * o.lambda_G_A_coh
            loss = loss + self.loss_G_A_coh
        cs = o.crop_size
        if self.aux['landmarks'] is not None:                                     # geometry loss :704-713
            mse = torch.nn.functional.mse_loss
            # both frames through the frozen regressor as ONE 2B batch (eval-mode BatchNorm: samples are independent)
            w4 = lambda w: torch.as_tensor(w).detach().to('cpu', torch.int32).reshape(-1, 4).expand(b, 4)     # noqa: E731
            lm12 = self.get_lm(torch.cat([self.fake_B, self.fake_B2], 0), torch.cat([w4(self.winB), w4(self.winB2)], 0))
            lm1, lm2 = lm12[:b], lm12[b:]
            t1, t2 = self.target_B_lm_68[:, :68], self.target_B2_lm_68[:, :68]
            if o.more_weight_for_lip != 2:
                g = mse(lm1 / cs, t1 / cs) + mse(lm2 / cs, t2 / cs)
            else:
                g = (mse(lm1[:, :48] / cs, t1[:, :48] / cs) + 2 * mse(lm1[:, 48:68] / cs, t1[:, 48:68] / cs)
                     + mse(lm2[:, :48] / cs, t2[:, :48] / cs) + 2 * mse(lm2[:, 48:68] / cs, t2[:, 48:68] / cs))
            self.loss_geom_B = g * o.lambda_geom
            loss = loss + self.loss_geom_B
        else:
            self._notice('landmarks', 'no landmark regressor: geometry loss (lambda_geom) is skipped')
        if o.lambda_geom_lipline > 0:                                             # :715-719
            m1, m2 = self.getlipline(self.target_B_lm_68), self.getlipline(self.target_B2_lm_68)
            self.loss_geom_B_lipline = (losses.weighted_mean(self.fake_B, m1, 1.0)
                                        + losses.weighted_mean(self.fake_B2, m2, 1.0)) * o.lambda_geom_lipline
            loss = loss + self.loss_geom_B_lipline
        if o.warp_loss:                                                           # :734-735
            self.loss_warp_B = losses.l1_loss(self.fake_B, self.fakeB_static_warp.detach(), o.lambda_warp)
            loss = loss + self.loss_warp_B
        rc = lambda lm: lm[:, :, [1, 0]]                                          # noqa: E731
        self.fake_B_warp = warp_nchw(self.fake_B.detach(), rc(self.target_B_lm_68), rc(self.target_B2_lm_68))  # :738
        self.loss_warp_inter1 = losses.l1_loss(self.fake_B2, self.fake_B_warp.detach(), o.lambda_warp_inter)
        loss = loss + self.loss_warp_inter1
        if self.aux['faceloss'] is not None and o.identity_loss in (1, 2):        # :741-752
            fl = self.aux['faceloss']
            # networks.FaceLoss reads a 1-channel drawing three times itself; other callables get the x3 repeat of :745-750
            rep = (lambda x: x) if isinstance(fl, networks.FaceLoss) else \
                (lambda x: x.repeat(1, 3, 1, 1) if x.shape[1] == 1 else x)         # noqa: E731
            if o.identity_loss == 1:
                gray = 0.299 * self.real_A[:, 0:1] + 0.587 * self.real_A[:, 1:2] + 0.114 * self.real_A[:, 2:3]
                other = rep(gray)
            else:
                other = rep(self.fakeB_static)
            self.loss_iden_B = torch.mean(fl(rep(self.fake_B), other, bbox1=self.winB, bbox2=self.winA)) * o.lambda_face
            loss = loss + self.loss_iden_B
        elif o.identity_loss:
            self._notice('faceloss', 'no face-recognition network: identity loss (lambda_face) is skipped')
        self.loss_G = loss
        with direct_param_grads():          # a plain .backward() of the train step: accumulate straight into the flat buffers
            loss.backward()

    # ------------------------------------------------------------------ step
    def optimize_parameters(self):
        """:782-819, with the two gradient all-reduces of the data-parallel step."""
        o = self.opt
        self.forward()
        nets_D = [getattr(self, 'net' + n) for n in self.model_names[1:]]
        self.set_requires_grad(nets_D, False)
        self.optimizer_G.zero_grad()
        self.backward_G()
        # G's gradients are exchanged right away (one RCCL all-reduce over xGMI); its update is deferred behind the D
        # backward passes, which read only the frames generated in forward() and the D weights.  With
        # parallel.OVERLAP_COLLECTIVES the exchange stays in flight under those passes; by default the compute stream
        # waits for it at once (kernels of another stream must not share the device with the matrix kernels, parallel.py)
        g_work = parallel.allreduce_optimizer_grads(self.optimizer_G, async_op=True)
        self.set_requires_grad(nets_D, True)
        self.optimizer_D.zero_grad()
        # every discriminator's gradients are exchanged as soon as its backward pass is enqueued (one collective per D
        # over its slice of the flat gradient buffer)
        d_works, d_whole = [], False

        def exchange(net):
            nonlocal d_whole
            w = parallel.allreduce_net_grads(self.optimizer_D, net)
            if w is False:
                d_whole = True              # no dense slice: one collective over the whole buffer at the end
            elif w is not None:
                d_works.append(w)
        self.backward_D_A()
        exchange(self.netD_A)
        if o.use_mask:
            self.backward_D_A_l()
            exchange(self.netD_A_l)
        if o.use_eye_mask:
            self.backward_D_A_le()
            exchange(self.netD_A_le)
        if o.use_lip_mask:
            self.backward_D_A_ll()
            exchange(self.netD_A_ll)
        if o.coherent:
            self.backward_D_A_coh()
            exchange(self.netD_A_coh)
        parallel.wait_work(g_work)
        self.optimizer_G.step()
        if d_whole:
            if d_works:
                raise RuntimeError('optimizer_D: some discriminators have a dense gradient slice and some do not')
            parallel.allreduce_optimizer_grads(self.optimizer_D)
        for w in d_works:
            parallel.wait_work(w)
        self.optimizer_D.step()
