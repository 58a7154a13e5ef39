#!/usr/bin/env python3
"""Literal fixtures for the two OpenCV rasterisation rules (tests/golden/opencv_rules.json), derived from the restatement of
OpenCV 4.2.0 modules/imgproc/src/drawing.cpp in oracle/cv_raster.py (Circle(), ThickLine(), FillConvexPoly(), Line2() --
the rules are quoted there).  cv2 is absent from the build image, so these are NOT outputs of cv2; a maintainer with
opencv-python==4.2.0.34 (requirements.txt:2) verifies them with  python tests/golden/check_opencv_rules.py .
Rasters are stored as rows of (y, x_first, x_last) runs on a 40 x 40 canvas."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

LINES = [  # (p0, p1, thickness): horizontal, vertical, 45 degrees, two general slopes, a thick one (512-px crops use 4)
    ((5, 8), (30, 8), 2), ((12, 4), (12, 33), 2), ((6, 6), (28, 28), 2), ((4, 30), (35, 11), 2), ((20, 3), (27, 36), 2),
    ((7, 9), (33, 25), 4), ((15, 15), (15, 15), 2)]


def runs(img):
    out = []
    for y in range(img.shape[0]):
        xs = np.flatnonzero(img[y])
        if xs.size:
            brk = np.flatnonzero(np.diff(xs) > 1)
            starts = np.concatenate(([0], brk + 1))
            ends = np.concatenate((brk, [xs.size - 1]))
            out += [[int(y), int(xs[a]), int(xs[b])] for a, b in zip(starts, ends)]
    return out


def main():
    from oracle import cv_raster as cr
    d = {'source': 'OpenCV 4.2.0 modules/imgproc/src/drawing.cpp: Circle() (cv::circle, thickness < 0), ThickLine() -> '
                   'FillConvexPoly(shift = 16) + Line2() + Circle() end caps (cv::line, thickness >= 2, LINE_8)',
         'canvas': 40,
         'circle_half_widths': {str(r): cr.circle_half_widths(r) for r in range(1, 9)},
         'lines': []}
    for p0, p1, th in LINES:
        img = np.zeros((40, 40), np.uint8)
        cr.thick_line(img, p0, p1, th)
        d['lines'].append({'p0': list(p0), 'p1': list(p1), 'thickness': th, 'runs': runs(img)})
    json.dump(d, open(os.path.join(HERE, 'opencv_rules.json'), 'w'), indent=0)
    print('wrote opencv_rules.json:', {k: v for k, v in d['circle_half_widths'].items()})


if __name__ == '__main__':
    main()
