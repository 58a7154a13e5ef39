"""Oracle (test infrastructure): the glue around the frozen auxiliary networks and the data-side rasterisers.

CPU restatements, plain PyTorch / numpy, of
* ``GeomGMIFWForeModel.get_lm``        Module2/models/geomgm_ifw_fore_model.py:390-415 (applied per sample)
* ``FaceLoss.crop_head_bbox/forward``  Module2/models/networks.py:2881-2966
* ``kp_to_map`` / ``flow_network_warp`` Module2/models/geomgm_ifw_fore_model.py:19-51, 69-84
* ``draw2(op=0)``                       Module2/data/umlvdfw_test_dataset.py:34-41 (cv2.circle, filled)

Pinning: the first three groups are pinned to outputs of the reference's own functions run in the build container
(``tests/golden/make_aux_golden.py`` -> ``tests/golden/aux.npz``; the frozen nets are replaced there by the fixed-seed
stand-ins of ``animateportrait_amd/standins.py`` because their checkpoints are not in the reference tree).
``draw2`` needs OpenCV, which is absent from this image: ``cv2_filled_circle_rows`` restates the octant walk of
``Circle()`` in OpenCV's modules/imgproc/src/drawing.cpp (opencv-python==4.2.0.34, requirements.txt:2) -- PARITY
UNPINNED against cv2 for that one function (no cv2 here to run it against); the row tables it yields (r = 3: 3/2/2/0,
r = 4: 4/3/3/2/0, ...) are committed as literals in tests/golden/opencv_rules.json next to the thick-line rasters of
oracle/cv_raster.py, and tests/golden/check_opencv_rules.py is the one-minute check for a maintainer with cv2.
"""
import numpy as np
import torch
import torch.nn.functional as F


def crop_box(x, win):
    """Ones-filled (x2-x1)^2 box with the window's part of image ``x`` (1, C, H, W) copied in (:392-401 / :2957-2961)."""
    _, c, h, w = x.shape
    x1, x2, y1, y2 = [int(v) for v in win]
    size = x2 - x1
    box = torch.ones((1, c, size, size), dtype=x.dtype)
    box[:, :, max(0, y1) - y1:min(y2, h) - y1, max(0, x1) - x1:min(w, x2) - x1] = \
        x[:, :, max(0, y1):min(y2, h), max(0, x1):min(w, x2)]
    return box


def get_lm_box(x, win, out_size=112):
    """The tensor ``get_lm`` feeds to the landmark net (:392-410): crop, BGR / x3, bicubic to out_size, [0, 1]."""
    out = []
    for i in range(x.shape[0]):
        box = crop_box(x[i:i + 1], win[i])
        box = box[:, [2, 1, 0]] if box.shape[1] == 3 else box.repeat(1, 3, 1, 1)
        box = F.interpolate(box, size=(out_size, out_size), mode='bicubic', align_corners=False)
        out.append((box + 1) * 0.5)
    return torch.cat(out, 0)


def get_lm(net, x, win, out_size=112):
    """:390-415 per sample: landmarks (N, 68, 2) in image pixels."""
    lm = net(get_lm_box(x, win, out_size))
    lm = (lm[0] if isinstance(lm, (tuple, list)) else lm).view(x.shape[0], 68, 2)
    w = torch.as_tensor(win).to(x.dtype).reshape(-1, 4)
    scale = torch.stack([w[:, 1] - w[:, 0], w[:, 3] - w[:, 2]], 1).view(-1, 1, 2)
    off = torch.stack([w[:, 0], w[:, 2]], 1).view(-1, 1, 2)
    return lm * scale + off


def crop_head_bbox(imgs, bboxs, height=112, width=96):
    """FaceLoss.crop_head_bbox (:2946-2966): 3-channel ones box, bilinear align_corners=True to height x width."""
    out = []
    for i in range(imgs.shape[0]):
        head = crop_box(imgs[i:i + 1], bboxs[i])
        out.append(F.interpolate(head, size=(height, width), mode='bilinear', align_corners=True))
    return torch.cat(out, 0)


def face_loss(net, imgs1, imgs2, bbox1, bbox2):
    """FaceLoss.forward with bboxes + compute_loss (:2881-2940): sum over the feature list of L1(f1, f2.detach())."""
    f1, f2 = net(crop_head_bbox(imgs1, bbox1)), net(crop_head_bbox(imgs2, bbox2))
    loss = 0.0
    for a, b in zip(f1, f2):
        loss = loss + F.l1_loss(a, b.detach())
    return loss


def kp_to_map(size, kps, radius=4):
    """kp_to_map((w, h), kps, 'binary', radius) (:19-44) for one sample: numpy, exactly the reference's expressions."""
    w, h = size
    x_grid, y_grid = np.meshgrid(range(w), range(h), indexing='xy')
    m = []
    for x, y in kps:
        if x == -1 or y == -1:
            m.append(np.zeros((h, w)).astype(np.float32))
        else:
            m.append(((x_grid - x) ** 2 + (y_grid - y) ** 2 <= radius ** 2).astype(np.float32))
    return torch.from_numpy(np.stack(m, axis=2).transpose((2, 0, 1)))


def kp_to_map_some(size, kps):
    return torch.stack([kp_to_map(size, k) for k in kps], 0)


def flow_network_warp(netF, real_A, lm1, lm2):
    """:69-84 on the CPU."""
    with torch.no_grad():
        j1 = kp_to_map_some((224, 224), lm1.cpu().numpy() * 7 / 8)
        j2 = kp_to_map_some((224, 224), lm2.cpu().numpy() * 7 / 8)
        jm = torch.cat([j1, j2], 1)                               # float32 joint maps whatever the model's dtype
        pdt = next((p.dtype for p in netF.parameters()), jm.dtype) if hasattr(netF, 'parameters') else jm.dtype
        flow_out, vis_out = netF(jm.to(pdt))[:2]
        vis = vis_out.argmax(dim=1, keepdim=True).float()
        mask = (vis < 2).float()
        flow = flow_out * 20. * mask
        s = real_A.shape[-1]
        warp_flow = F.interpolate(flow / 7 * 8, size=(s, s), mode='bilinear', align_corners=True)
        res_mask = F.interpolate(mask, size=(s, s), mode='bilinear', align_corners=True)
    return warp_flow, res_mask


def cv2_filled_circle_rows(radius):
    """Half width per |dy| row of cv2.circle(img, c, radius, color, -1): the octant walk of drawing.cpp Circle()."""
    hw = [-1] * (radius + 1)
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        hw[dy] = max(hw[dy], dx)
        hw[dx] = max(hw[dx], dy)
        dy += 1
        err += plus
        plus += 2
        if err > 0:
            err -= minus
            dx -= 1
            minus -= 2
    return hw


def draw2(height, width, lands, radius=3):
    """draw2(height, width, lands, radius, thickness, op=0) (:34-41) -> (1, height, width) in {-1, +1}."""
    frame = np.zeros((height, width), dtype=np.uint8)
    hw = cv2_filled_circle_rows(radius)
    for x, y in np.round(np.asarray(lands)).astype(int):
        for dy in range(-radius, radius + 1):
            yy = y + dy
            if 0 <= yy < height:
                lo, hi = max(0, x - hw[abs(dy)]), min(width - 1, x + hw[abs(dy)])
                if lo <= hi:
                    frame[yy, lo:hi + 1] = 255
    return torch.from_numpy(frame)[None, ...].float() / 255. * 2 - 1
