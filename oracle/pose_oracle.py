"""CPU restatement (numpy) of the OpenPose control-map drawing spec of include/gip_pose.h — TEST INFRASTRUCTURE ONLY.

Follows threestudio/utils/poser.py:832-904 (Skeleton.openpose_draw): discs in key-point order, then limbs in line order,
each limb blended 0.4 / 0.6 into the canvas with uint8 rounding.  Shapes follow the published OpenCV algorithms
(midpoint circle of radius 4; ellipse of half-axes (int(len/2), 4)); cv2 is not installed here, so the footprint of a
limb is the analytic ellipse inflated by half a pixel rather than cv2's polygon scan conversion: PARITY AGAINST OPENCV
IS UNPINNED (boundary pixels may differ); the HIP kernel is pinned bit-exactly against THIS file.
Only tests/ may import this module; the product path never does."""
import numpy as np

COLORS = np.array([[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0],
                   [0, 255, 85], [0, 255, 170], [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255],
                   [170, 0, 255], [255, 0, 255], [255, 0, 170], [255, 0, 85]], np.uint8)      # poser.py:701-703
LINES = np.array([[0, 1], [1, 2], [2, 3], [3, 4], [1, 5], [5, 6], [6, 7], [1, 8], [8, 9], [9, 10], [1, 11], [11, 12],
                  [12, 13], [0, 14], [14, 16], [0, 15], [15, 17]], np.int32)                    # poser.py:688
DISC_HALF = np.array([4, 3, 3, 2, 0], np.int32)   # cv2.circle(r=4, filled): half-width per |dy| of the midpoint circle


def draw(points_px, visible, limbs, H, W):
    """points_px [18,2] int, visible [18] bool, limbs [17,6] float32 (cx, cy, a, on, cos, sin) -> [H,W,3] float32."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.int32), np.arange(W, dtype=np.int32), indexing="ij")
    canvas = np.zeros((H, W, 3), np.uint8)
    for i in range(18):                                                           # poser.py:879-882
        if not visible[i]:
            continue
        dy, dx = np.abs(yy - points_px[i, 1]), np.abs(xx - points_px[i, 0])
        inside = (dy <= 4) & (dx <= DISC_HALF[np.minimum(dy, 4)])
        canvas[inside] = COLORS[i]
    for l in range(17):                                                           # poser.py:885-899
        cx, cy, a, on, cs, sn = [np.float32(v) for v in limbs[l]]
        if on == 0:
            continue
        dx, dy = (xx - np.int32(cx)).astype(np.float32), (yy - np.int32(cy)).astype(np.float32)
        u = dx * cs + dy * sn
        w = -dx * sn + dy * cs
        ua = u / np.float32(np.float32(np.int32(a)) + np.float32(0.5))
        wb = w / np.float32(4.5)
        inside = (ua * ua + wb * wb) <= np.float32(1.0)
        src = np.where(inside[..., None], COLORS[l][None, None, :], canvas)
        t = canvas.astype(np.float32) * np.float32(0.4) + src.astype(np.float32) * np.float32(0.6)
        canvas = np.clip(np.rint(t), 0, 255).astype(np.uint8)                    # cv2.addWeighted: round half to even
    return canvas.astype(np.float32) / np.float32(255.0)
