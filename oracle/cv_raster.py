"""Oracle (test infrastructure): the two OpenCV rasterisation rules the reference's data / loss code relies on, restated
from OpenCV's source because cv2 cannot be imported in the build image.  **Parity unpinned against cv2 itself** -- the
literal fixtures derived from this file (tests/golden/opencv_rules.json) come with a checker a maintainer with
opencv-python installed runs in one minute:  python tests/golden/check_opencv_rules.py

Reference call sites
  * ``cv2.circle(frame, (x, y), radius, 255, -1)``   Module2/data/umlvdfw_test_dataset.py:34-41 (``draw2`` op = 0),
  * ``cv2.line(mask, p0, p1, 255, thickness)``        Module2/models/geomgm_ifw_fore_model.py:507-515 (``getlipline``;
    thickness 2 at 256 px, 4 at 512 px; float landmark coordinates are truncated to int by the Python binding).
OpenCV source restated (opencv-python==4.2.0.34 -> OpenCV 4.2.0, modules/imgproc/src/drawing.cpp):
  * ``cv::circle``: thickness < 0, LINE_8, shift 0  ->  ``Circle(img, center, radius, color, fill = 1)``: the octant walk
    ``err = 0, dx = radius, dy = 0, plus = 1, minus = 2 radius - 1``; per step the rows ``cy +- dy`` are filled over
    ``cx +- dx`` and the rows ``cy +- dx`` over ``cx +- dy``; then ``dy++, err += plus, plus += 2`` and, when
    ``err > 0``: ``err -= minus, dx--, minus -= 2``; loop while ``dx >= dy``.
  * ``cv::line`` -> ``ThickLine(img, p0, p1, color, thickness, LINE_8, flags = 3, shift = 0)``; for thickness > 1 (16.16
    fixed point, XY_SHIFT = 16): ``dp = (round(dy r), round(dx r))`` with ``dx = p0.x - p1.x, dy = p1.y - p0.y`` and
    ``r = (thickness << 15 + odd * 32768) / sqrt(dx^2 + dy^2)``; the quad ``p0 + dp, p0 - dp, p1 - dp, p1 + dp`` goes to
    ``FillConvexPoly(..., shift = XY_SHIFT)`` (outline by ``Line2``, interior by scanlines with
    ``dx_edge = ((xe - xs) 2 + (ty - y)) / (2 (ty - y))``, ``xx1 = (x_left + 32768) >> 16``, ``xx2 = (x_right + 32768) >> 16``),
    and both end points get ``Circle(round(p), (thickness_fixed + 32768) >> 16, fill)``.
"""
import numpy as np

XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT


def circle_half_widths(radius):
    """Half width of the filled row at |dy| = 0 .. radius of ``Circle(..., fill)``."""
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


def fill_circle(img, cx, cy, radius, value=255):
    h, w = img.shape
    for dy, half in enumerate(circle_half_widths(radius)):
        for yy in {cy - dy, cy + dy}:
            if 0 <= yy < h:
                lo, hi = max(0, cx - half), min(w - 1, cx + half)
                if lo <= hi:
                    img[yy, lo:hi + 1] = value
    return img


def _trunc_div(a, b):
    """C integer division (truncation toward zero) on python ints."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def _line2(img, p1, p2, value):
    """``Line2``: DDA over the major axis in 16.16 fixed point (no clipping needed: points are put only when inside)."""
    h, w = img.shape
    x1, y1 = p1
    x2, y2 = p2
    dx, dy = x2 - x1, y2 - y1
    ax, ay = abs(dx), abs(dy)

    def put(x, y):
        if 0 <= x < w and 0 <= y < h:
            img[y, x] = value
    if ax > ay:
        if dx < 0:
            x1, y1, x2, y2, dy = x2, y2, x1, y1, -dy
        y_step = _trunc_div(dy << XY_SHIFT, ax | 1)
        ecount = (x2 - x1) >> XY_SHIFT
        put((x2 + (XY_ONE >> 1)) >> XY_SHIFT, (y2 + (XY_ONE >> 1)) >> XY_SHIFT)
        x, y = (x1 + (XY_ONE >> 1)) >> XY_SHIFT, y1 + (XY_ONE >> 1)
        while ecount >= 0:
            put(x, y >> XY_SHIFT)
            x += 1
            y += y_step
            ecount -= 1
    else:
        if dy < 0:
            x1, y1, x2, y2, dx = x2, y2, x1, y1, -dx
        x_step = _trunc_div(dx << XY_SHIFT, ay | 1)
        ecount = (y2 - y1) >> XY_SHIFT
        put((x2 + (XY_ONE >> 1)) >> XY_SHIFT, (y2 + (XY_ONE >> 1)) >> XY_SHIFT)
        x, y = x1 + (XY_ONE >> 1), (y1 + (XY_ONE >> 1)) >> XY_SHIFT
        while ecount >= 0:
            put(x >> XY_SHIFT, y)
            x += x_step
            y += 1
            ecount -= 1


def _fill_convex_poly(img, v, value):
    """``FillConvexPoly(img, v, npts, color, LINE_8, shift = XY_SHIFT)`` for fixed-point vertices v (list of (x, y))."""
    h, w = img.shape
    npts = len(v)
    delta = XY_ONE >> 1
    p0 = v[-1]
    imin, ymin, ymax = 0, v[0][1], v[0][1]
    xmin = xmax = v[0][0]
    for i, p in enumerate(v):
        if p[1] < ymin:
            ymin, imin = p[1], i
        ymax = max(ymax, p[1])
        xmax, xmin = max(xmax, p[0]), min(xmin, p[0])
        _line2(img, p0, p, value)
        p0 = p
    xmin, xmax = (xmin + delta) >> XY_SHIFT, (xmax + delta) >> XY_SHIFT
    ymin, ymax = (ymin + delta) >> XY_SHIFT, (ymax + delta) >> XY_SHIFT
    if npts < 3 or xmax < 0 or ymax < 0 or xmin >= w or ymin >= h:
        return
    ymax = min(ymax, h - 1)
    edge = [dict(idx=imin, di=1, x=-XY_ONE, dx=0, ye=ymin), dict(idx=imin, di=npts - 1, x=-XY_ONE, dx=0, ye=ymin)]
    edges = npts
    y = ymin
    while True:
        for e in edge:
            if y >= e['ye']:
                idx0, di = e['idx'], e['di']
                idx = (idx0 + di) % npts
                while True:
                    edges -= 1
                    if edges < 0:
                        break
                    ty = (v[idx][1] + delta) >> XY_SHIFT
                    if ty > y:
                        xs, xe = v[idx0][0], v[idx][0]
                        e['ye'] = ty
                        e['dx'] = _trunc_div((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e['x'] = xs
                        e['idx'] = idx
                        break
                    idx0 = idx
                    idx = (idx + di) % npts
        if edges < 0:
            break
        if y >= 0:
            l, r = (edge[0], edge[1]) if edge[0]['x'] <= edge[1]['x'] else (edge[1], edge[0])
            xx1 = (l['x'] + delta) >> XY_SHIFT
            xx2 = (r['x'] + delta) >> XY_SHIFT
            if xx2 >= 0 and xx1 < w:
                img[y, max(xx1, 0):min(xx2, w - 1) + 1] = value
        edge[0]['x'] += edge[0]['dx']
        edge[1]['x'] += edge[1]['dx']
        y += 1
        if y > ymax:
            break


def _cv_round(x):
    return int(np.rint(x))           # cvRound: round half to even (lrint)


def thick_line(img, p0, p1, thickness, value=255):
    """``cv2.line(img, p0, p1, value, thickness)`` for thickness >= 2 and integer end points (LINE_8, shift 0)."""
    x0, y0 = int(p0[0]) << XY_SHIFT, int(p0[1]) << XY_SHIFT
    x1, y1 = int(p1[0]) << XY_SHIFT, int(p1[1]) << XY_SHIFT
    dx, dy = (x0 - x1) / XY_ONE, (y1 - y0) / XY_ONE
    r = dx * dx + dy * dy
    odd = thickness & 1
    tf = thickness << (XY_SHIFT - 1)
    if abs(r) > np.finfo(np.float64).eps:
        r = (tf + odd * XY_ONE * 0.5) / np.sqrt(r)
        dpx, dpy = _cv_round(dy * r), _cv_round(dx * r)
        _fill_convex_poly(img, [(x0 + dpx, y0 + dpy), (x0 - dpx, y0 - dpy), (x1 - dpx, y1 - dpy), (x1 + dpx, y1 + dpy)], value)
    rad = (tf + (XY_ONE >> 1)) >> XY_SHIFT
    for (px, py) in ((x0, y0), (x1, y1)):
        fill_circle(img, (px + (XY_ONE >> 1)) >> XY_SHIFT, (py + (XY_ONE >> 1)) >> XY_SHIFT, rad, value)
    return img


def lip_line_mask(size, lands, segments, thickness):
    """``getlipline`` (geomgm_ifw_fore_model.py:507-515) for one sample: lands (68, 2) float (x, y) -> (size, size) in {0, 1}."""
    img = np.zeros((size, size), dtype=np.uint8)
    for a, b in segments:
        thick_line(img, (int(lands[a][0]), int(lands[a][1])), (int(lands[b][0]), int(lands[b][1])), thickness)
    return (img > 0).astype(np.float32)
