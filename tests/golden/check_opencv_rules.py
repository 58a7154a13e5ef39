#!/usr/bin/env python3
"""One-minute check for a maintainer WITH opencv-python installed (it is absent from the build image): do the literal
fixtures of tests/golden/opencv_rules.json equal what cv2.circle / cv2.line really draw?

    pip install opencv-python==4.2.0.34 && python tests/golden/check_opencv_rules.py
"""
import json
import os

import numpy as np


def main():
    import cv2
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'opencv_rules.json')))
    n = d['canvas']
    bad = 0
    for r, hw in d['circle_half_widths'].items():
        r = int(r)
        img = np.zeros((n, n), np.uint8)
        cv2.circle(img, (20, 20), r, 255, -1)
        got = [int((img[20 + dy] > 0).sum() - 1) // 2 if (img[20 + dy] > 0).any() else -1 for dy in range(r + 1)]
        ok = got == hw
        bad += not ok
        print('circle r=%d: fixture %s cv2 %s %s' % (r, hw, got, 'OK' if ok else 'MISMATCH'))
    for ln in d['lines']:
        img = np.zeros((n, n), np.uint8)
        cv2.line(img, tuple(ln['p0']), tuple(ln['p1']), 255, ln['thickness'])
        want = np.zeros((n, n), np.uint8)
        for y, a, b in ln['runs']:
            want[y, a:b + 1] = 255
        diff = int((img != want).sum())
        bad += diff > 0
        print('line %s -> %s thickness %d: %d pixels differ %s' % (ln['p0'], ln['p1'], ln['thickness'], diff, 'OK' if diff == 0 else 'MISMATCH'))
    raise SystemExit(1 if bad else 0)


if __name__ == '__main__':
    main()
