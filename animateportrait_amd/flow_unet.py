"""The intrinsic-flow regressor ``netF`` of the models as a loadable module.

``FlowUnetV2`` mirrors ``FlowUnet_v2`` (Module2/intrinsic_flow_models/networks.py:647-744; its ``ResidualBlock`` :26-60
and the ``conv`` / ``channel_mapping`` helpers :16-24) attribute for attribute, so the reference's checkpoint
(``checkpoints/FlowReg_id_flow_faces/best_net_netF.pth``, flow_regression_model.py:56-60, base_model.py:59-71) loads with
``strict=True``.  It is a frozen third-party network: stock PyTorch-ROCm modules, no hand kernels (SURVEY.md section 2
row 12).  Its hyper-parameters are not in the reference tree -- they live in the checkpoint-side ``train_opt.json`` that
``load_flow_network`` reads (geomgm_ifw_fore_model.py:57-68) -- so they are constructor arguments here and
``load_flow_network`` takes them from the same file.  The stages around the network (136 joint maps in, masked /
rescaled / resized flow out) are the device kernels of ``losses.flow_network_warp``; attach the module with
``model.aux['netF'] = load_flow_network(...)``.
"""
import json
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv(cin, cout, kernel_size, padding, norm_layer, bias):                   # networks.py:16-21
    return nn.Sequential(nn.Conv2d(cin, cout, kernel_size, 1, padding, 1, bias=bias), norm_layer(cout))


class ResidualBlock(nn.Module):
    """networks.py:26-60 (the two forms FlowUnet_v2 builds: without / with an additional input)."""

    def __init__(self, dim, dim_a, norm_layer, use_bias, activation):
        super().__init__()
        self.activation = activation
        if dim_a is None or dim_a <= 0:
            self.conv = _conv(dim, dim, 3, 1, norm_layer, use_bias)
        else:
            self.conv_a = _conv(dim_a, dim, 1, 0, norm_layer, use_bias)        # channel_mapping :23-24
            self.conv = _conv(dim * 2, dim, 3, 1, norm_layer, use_bias)

    def forward(self, x, a=None):
        residual = x if a is None else torch.cat((x, self.conv_a(self.activation(a))), dim=1)
        return x + self.conv(self.activation(residual))


class FlowUnetV2(nn.Module):
    def __init__(self, input_nc, nf=64, max_nf=256, start_scale=2, num_scales=7, n_residual_blocks=2, norm='batch'):
        super().__init__()
        self.start_scale, self.num_scales, self.n_residual_blocks = start_scale, num_scales, n_residual_blocks
        if norm == 'batch':
            norm_layer, use_bias = nn.BatchNorm2d, False
        elif norm == 'instance':
            norm_layer, use_bias = nn.InstanceNorm2d, True
        else:
            raise NotImplementedError('norm [%s]' % norm)
        activation = nn.ReLU(False)
        start_level = int(np.log2(start_scale))
        pre_conv = [_conv(input_nc, nf, 1, 0, norm_layer, use_bias)]
        for i in range(start_level):
            c_in, c_out = min(nf * (i + 1), max_nf), min(nf * (i + 2), max_nf)
            pre_conv += [ResidualBlock(c_in, None, norm_layer, use_bias, activation), activation,
                         nn.Conv2d(c_in, c_out, kernel_size=3, stride=2, padding=1, bias=use_bias), norm_layer(c_out)]
        self.pre_conv = nn.Sequential(*pre_conv)
        for l in range(num_scales):
            c_in, c_out = min(nf * (start_level + l + 1), max_nf), min(nf * (start_level + l + 2), max_nf)
            for i in range(n_residual_blocks):
                setattr(self, 'enc_%d_res_%d' % (l, i), ResidualBlock(c_in, None, norm_layer, use_bias, activation))
            setattr(self, 'enc_%d_downsample' % l, nn.Sequential(
                activation, nn.Conv2d(c_in, c_out, kernel_size=3, stride=2, padding=1, bias=use_bias), norm_layer(c_out)))
            setattr(self, 'dec_%d_upsample' % l, nn.Sequential(
                activation, nn.Conv2d(c_out, c_in * 4, kernel_size=3, padding=1, bias=use_bias), nn.PixelShuffle(2),
                norm_layer(c_in)))
            for i in range(n_residual_blocks):
                setattr(self, 'dec_%d_res_%d' % (l, i), ResidualBlock(c_in, c_in, norm_layer, use_bias, activation))
            setattr(self, 'pred_flow_%d' % l, nn.Sequential(activation, nn.Conv2d(c_in, 2, kernel_size=3, padding=1, bias=True)))
        self.pred_vis = nn.Sequential(activation, nn.Conv2d(nf * (1 + start_level), 3, kernel_size=3, padding=1, bias=True))

    def forward(self, x):                                                        # :718-744
        hiddens, flow_pyr = [], []
        x = self.pre_conv(x)
        for l in range(self.num_scales):
            for i in range(self.n_residual_blocks):
                x = getattr(self, 'enc_%d_res_%d' % (l, i))(x)
                hiddens.append(x)
            x = getattr(self, 'enc_%d_downsample' % l)(x)
        for l in range(self.num_scales - 1, -1, -1):
            x = getattr(self, 'dec_%d_upsample' % l)(x)
            for i in range(self.n_residual_blocks - 1, -1, -1):
                x = getattr(self, 'dec_%d_res_%d' % (l, i))(x, hiddens.pop())
            flow_pyr = [getattr(self, 'pred_flow_%d' % l)(x)] + flow_pyr
        up = lambda t: F.interpolate(t, scale_factor=self.start_scale, mode='bilinear', align_corners=False)   # noqa: E731
        return up(flow_pyr[0]), up(self.pred_vis(x)), flow_pyr, x


def input_dim(opt, input_type):
    """FlowRegressionModel.get_input_dim (flow_regression_model.py:159-178) for the items a flow regressor is fed."""
    dims = {'img': 3, 'seg': opt.get('seg_nc', 0), 'joint': opt.get('joint_nc', 0), 'flow': 2, 'flow_gt': 2, 'vis': 3}
    return sum(dims[item] for item in sorted(input_type.split('+')))


def load_flow_network(model_id='FlowReg_id_flow_faces', epoch='best', checkpoints_dir='checkpoints', device=None):
    """load_flow_network (geomgm_ifw_fore_model.py:57-68): hyper-parameters from ``<dir>/<model_id>/train_opt.json``,
    weights from ``<epoch>_net_netF.pth``; returns the module in eval mode."""
    d = os.path.join(checkpoints_dir, model_id)
    opt = json.load(open(os.path.join(d, 'train_opt.json')))
    if opt.get('which_model', 'unet_v2') != 'unet_v2':
        raise NotImplementedError('flow network [%s]: only FlowUnet_v2 is mirrored' % opt.get('which_model'))
    net = FlowUnetV2(input_dim(opt, opt['input_type1']) + input_dim(opt, opt['input_type2']), nf=opt['nf'],
                     max_nf=opt['max_nf'], start_scale=opt['start_scale'], num_scales=opt['num_scale'], norm=opt['norm'])
    net.load_state_dict(torch.load(os.path.join(d, '%s_net_netF.pth' % epoch), map_location='cpu'), strict=True)
    net.eval()
    for p in net.parameters():
        p.requires_grad_(False)
    return net.to(device) if device is not None else net
