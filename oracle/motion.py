"""Oracle (test infrastructure): the data layer's landmark -> motion grid map.

Restatement of ``cal_motion256`` (Module2/data/umlvd_ifw_dataset.py:60-74 == umlvdfw_test_dataset.py:67-81): the source
landmark positions are interpolated linearly over the Delaunay triangulation of the destination landmarks plus 8 border
points (``scipy.interpolate.griddata(method='linear')``, scipy 1.15 in this image -- the reference's own dependency) at
every pixel of the 256 x 256 grid, and normalised to grid_sample coordinates.  Pinned to the output of the reference's
function in tests/golden/motion.npz (tests/golden/make_motion_golden.py).
"""
import numpy as np


def cal_motion256(lm2d0, lm2d):
    """lm2d0, lm2d: (68, 2) arrays of (x, y) pixels.  Returns (256, 256, 2) float32, [..., 0] = x, [..., 1] = y."""
    from scipy.interpolate import griddata
    grid_x, grid_y = np.mgrid[0:255:256j, 0:255:256j]
    edges = np.array([[0, 0], [255, 255], [0, 255], [255, 0], [0, 255], [255, 0], [255, 255], [255, 255]])
    lm2d = np.asarray(lm2d)[:, [1, 0]]
    lm2d0 = np.asarray(lm2d0)[:, [1, 0]]
    destination = np.concatenate((lm2d, edges))
    source = np.concatenate((lm2d0, edges))
    grid_z = griddata(destination, source, (grid_x, grid_y), method='linear')      # (256, 256, 2) as (row, col)
    map_xy = np.stack([grid_z[..., 1].astype('float32'), grid_z[..., 0].astype('float32')], axis=2)
    return map_xy / 127.5 - 1
