"""``--model geomcgt_ifw_test`` on the MI355X: the streaming-inference model of
Module2/models/geomcgt_ifw_test_model.py (options :178-189, __init__ :191-229, set_input :254-274,
forward :276-302), the caller that ``main_end2end_module2.py:96-97`` runs once per video frame.

What is the same: option names / defaults, ``netG_A`` through ``networks.define_G`` with ``--netg_resb_div/disp``, the
static drawing generator ``define_G(3, 1, 64, 'resnet_style2_9blocks', 'instance')`` for experiment names that contain
'drawing', the visual names, and every formula of ``forward`` (matte threshold, foreground compositing of the photo,
512^2 static drawing resized to 256^2, hot-path generator, ``grid_sample(mask, warp_motion, align_corners=True)``
blend of the animated foreground over the static drawing).

What is different (and why):
* The frozen third-party nets have no checkpoints in the reference tree: MODNet (matte)
  and netF (intrinsic flow) enter through ``self.aux`` callables; absent, their outputs are read from the input
  dict (``matte`` or ``mask``, ``iw_flow`` / ``if_mask``), as in the training model.
* The static drawing depends on the photo only, and the photo is constant over a clip: it is computed once per
  distinct ``real_A`` tensor and cached, where the reference recomputes it per frame.
* Batched: every formula is applied per sample (the reference is written for batch size 1).
* The 'cartoon' branch needs the third-party Photo2Cartoon checkpoint and is outside the path.
"""
import torch

from .. import losses, networks, ops
from .base_model import BaseModel


class GeomCGTIFWTestModel(BaseModel):
    @staticmethod
    def modify_commandline_options(parser, is_train=True):                          # :178-189
        parser.set_defaults(no_dropout=True)
        parser.add_argument('--netg_resb_div', type=int, default=3, help='div')
        parser.add_argument('--netg_resb_disp', type=int, default=1, help='disp')
        parser.add_argument('--truncate', type=float, default=0.0, help='whether truncate in forward')
        parser.set_defaults(dataset_mode='umlvdfw_test')
        parser.add_argument('--draw_op', type=int, default=0, help='use which format to draw landmark')
        parser.add_argument('--blendbg', type=int, default=0, help='whether blend with bg')
        return parser

    def __init__(self, opt):
        BaseModel.__init__(self, opt)
        self.visual_names = ['real_A', 'real_A_lm', 'target_B_lm', 'fake_B', 'fake_B_vis', 'fg_mask', 'fakeB_static',
                             'fake_B_fore', 'fg_mask1']                            # :199-204
        self.model_names = ['G_A']
        gid = [self.gpu_ids[0]]
        self.netG_A = networks.define_G(opt.input_nc, opt.output_nc, opt.ngf, opt.netG, opt.norm, not opt.no_dropout,
                                        opt.init_type, opt.init_gain, gid, div=opt.netg_resb_div,
                                        disp=opt.netg_resb_disp)                   # :211-213
        self.aux = {'modnet': None, 'netF': None}
        if 'cartoon' in opt.name:
            raise NotImplementedError('the cartoon branch needs the third-party Photo2Cartoon checkpoint '
                                      '(geomcgt_ifw_test_model.py:228-229) and is outside the MI355X hot path')
        # 'drawing' (the README / main_end2end configuration): static generator, frozen           :224-227
        self.net_staticG = networks.define_G(3, 1, 64, 'resnet_style2_9blocks', 'instance', use_dropout=False,
                                             gpu_ids=gid)
        self.net_staticG.eval()
        self._static_key = None
        self._style = None
        self._static_loaded = False

    STATIC_CHECKPOINT = 'checkpoints/static/drawing.pth'                              # :226

    def setup(self, opt):
        """base_model.py:79-90 plus the static generator's checkpoint, which the reference loads unconditionally in
        __init__ (:226).  A missing file is an error unless --allow_random_init (smoke mode) is set."""
        if not self._static_loaded:
            import os
            if os.path.exists(self.STATIC_CHECKPOINT):
                self.load_static(self.STATIC_CHECKPOINT)
            elif not getattr(opt, 'allow_random_init', False):
                raise FileNotFoundError('%s (static drawing generator) not found; load it with load_static(path) '
                                        'before setup() or pass --allow_random_init' % self.STATIC_CHECKPOINT)
            else:
                print('WARNING: --allow_random_init: static drawing generator runs with RANDOM weights')
        BaseModel.setup(self, opt)

    def load_static(self, path):
        """checkpoints/static/drawing.pth (:226): same key names, strict."""
        sd = torch.load(path, map_location=self.device)
        self.net_staticG.load_state_dict(sd, strict=True)
        self._static_key = None
        self._static_loaded = True

    # ------------------------------------------------------------------------------------------------ input
    def set_input(self, input):                                                      # :254-274
        AtoB = self.opt.direction == 'AtoB'
        dev = self.device
        photo = input['A' if AtoB else 'B']
        self._photo_key = (id(photo), photo._version, tuple(photo.shape))   # a clip passes the same photo tensor per frame
        self._photo_ref = photo                                             # (kept alive so that its id stays unique)
        self.real_A = photo.to(dev)
        self.warp_motion = input['warp_motion'].to(dev)
        alm = input['A_lm']
        self.real_A_lm = alm.to(dev)
        # The photo's landmark map is constant over a clip.  A streaming caller that guarantees it (ClipStreamer: one tensor
        # per batch size, never written to) names the clip with ``input['clip_id']``; its encoding inside the generator is then
        # cached under (clip id, shape, arithmetic mode).  Without a clip id -- dataset runs, bench steps -- there is no cache:
        # both landmark maps run through the encoder as one 2B pass.
        cid = input.get('clip_id')
        self.netG_A.land1_cache_key = None if cid is None else (cid, tuple(alm.shape), ops.DEFAULT_PRECISION)
        self.target_B_lm = input['tB_lm'].to(dev)
        for k, attr in (('realA_static_warp', 'realA_static_warp'), ('A_lm_68', 'real_A_lm_68'),
                        ('tB_lm_68', 'target_B_lm_68'), ('winB', 'winB')):
            if k in input:
                setattr(self, attr, input[k].to(dev) if torch.is_tensor(input[k]) else input[k])
        self.image_paths = input.get('image_paths')
        if self.aux['netF'] is not None:
            self.iw_flow, self.real_A_if_mask = losses.flow_network_warp(                       # :62-76, :270-271
                self.aux['netF'], self.real_A, self.real_A_lm_68[:, :68], self.target_B_lm_68[:, :68])
        else:
            self.iw_flow, self.real_A_if_mask = input['iw_flow'].to(dev), input['if_mask'].to(dev)
        self._matte_in = None
        if self.aux['modnet'] is None:
            self._matte_in = (input['matte'] if 'matte' in input else input['mask']).to(dev)

    # ------------------------------------------------------------------------------------------------ forward
    def static_drawing(self, real_A):
        """:280-285, cached per photo: style planes (0, 1, 0) at 128^2, 256 -> 512 -> G_static -> 256."""
        key = self._photo_key
        if self._static_key != key:
            n = real_A.shape[0]
            if self._style is None or self._style.shape[0] != n:
                self._style = torch.tensor([0., 1., 0.], device=real_A.device).view(1, 3, 1, 1).repeat(n, 1, 128, 128)
            a512 = ops.resize_bilinear(real_A.contiguous(), (512, 512))
            y512 = self.net_staticG(a512, self._style)
            self.fakeB_static = ops.resize_bilinear(y512, (256, 256))
            self._static_key = key
        return self.fakeB_static

    def forward(self):                                                               # :276-302
        if self.aux['modnet'] is not None:
            _, _, matte = self.aux['modnet'](self.real_A, True)
        else:
            matte = self._matte_in
        mask = (matte > 0.5).float()
        fakeB_static = self.static_drawing(self.real_A)       # of the unmasked photo, as :282-285
        self.real_A = losses.fore_composite(self.real_A, mask)
        self.fg_mask = (mask * 2 - 1).repeat(1, 3, 1, 1)
        with torch.no_grad():
            self.fake_B = self.netG_A(self.real_A.contiguous(), self.real_A_lm, self.target_B_lm, self.warp_motion,
                                      self.iw_flow, self.real_A_if_mask)
        self.mask1 = ops.grid_sample(mask.contiguous(), self.warp_motion.contiguous(), align_corners=True)
        self.fake_B_fore = self.fake_B.clone()
        self.fake_B = losses.bg_blend(self.fake_B, fakeB_static, self.mask1)
        self.fg_mask1 = (self.mask1 * 2 - 1).repeat(1, 3, 1, 1)
        if hasattr(self, 'target_B_lm_68') and hasattr(self, 'winB'):
            self.fake_B_vis = self.get_lmvis(self.fake_B, self.target_B_lm_68, self.winB)

    def get_lmvis(self, tensor_im, lm, win, hradius=3):                             # :232-251 (sample 0, as there)
        vis = tensor_im.detach().clone()
        if vis.shape[1] == 1:
            vis = vis.repeat(1, 3, 1, 1)
        pts = lm.detach().cpu().numpy()
        win = win.cpu().numpy() if torch.is_tensor(win) else win

        def mark(y0, y1, x0, x1):
            vis[:, 0, y0:y1, x0:x1] = 1
            vis[:, 1:, y0:y1, x0:x1] = -1
        for k in range(lm.shape[1]):
            x, y = int(round(float(pts[0, k, 0]))), int(round(float(pts[0, k, 1])))
            mark(y - hradius, y + hradius, x - hradius, x + hradius)
        x1, x2, y1, y2 = (int(win[0][i]) for i in range(4))
        mark(y1 - hradius, y1 + hradius, x1 - hradius, x2 + hradius)
        mark(y2 - hradius, y2 + hradius, x1 - hradius, x2 + hradius)
        mark(y1 - hradius, y2 + hradius, x1 - hradius, x1 + hradius)
        mark(y1 - hradius, y2 + hradius, x2 - hradius, x2 + hradius)
        return vis

    def optimize_parameters(self):
        pass                                                                         # test-time model
