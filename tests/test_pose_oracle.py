"""oracle/pose_oracle.py (the restatement of OpenCV's ellipse2Poly / FillConvexPoly / Line / Circle that the pose-map
kernel is pinned to) against properties that can be checked by hand — OpenCV itself is not installed here."""
import numpy as np

from oracle import pose_oracle as po


def test_disc_is_the_radius_4_midpoint_circle():
    m = po.disc_mask(21, 21, 10, 10)
    half = [int(m[10 + dy].sum() - 1) // 2 for dy in range(-5, 6)]
    assert half == [-1, 0, 2, 3, 3, 4, 3, 3, 2, 0, -1]           # rows |dy| = 0..4: half-widths 4, 3, 3, 2, 0 (cv2.circle, filled)
    assert m[10, 6:15].all() and not m[10, 5] and m.sum() == 9 + 2 * (7 + 7 + 5 + 1)
    # clipped at the border: same rows, cut
    c = po.disc_mask(21, 21, 1, 0)
    assert c[0, 0:6].all() and not c[0, 6] and c[4, 1] and c.sum() == 6 + 5 + 5 + 4 + 1


def test_sin_table_and_polygon():
    assert po.SIN_TABLE.shape == (451,) and po.SIN_TABLE[0] == 0 and po.SIN_TABLE[90] == 1 and po.SIN_TABLE[450] == 1
    assert abs(float(po.SIN_TABLE[1]) - 0.0174524) < 1e-9 and abs(float(po.SIN_TABLE[30]) - 0.5) < 1e-9
    pts = po.ellipse2poly(100, 80, 40, 4, 0)
    assert pts[0] == (140, 80) and pts[-1] == (140, 80)          # 0 and 360 degrees: the polygon closes on its first point
    s = set(pts)
    assert all((200 - x, 160 - y) in s for x, y in s)            # point symmetry about the centre
    assert max(x for x, _ in pts) == 140 and min(x for x, _ in pts) == 60 and max(y for _, y in pts) == 84 and min(y for _, y in pts) == 76
    assert all(a != b for a, b in zip(pts, pts[1:]))             # consecutive duplicates removed
    # negative angles are normalised by +360 (poser.py passes int(degrees(atan2)) in [-180, 180])
    assert po.ellipse2poly(50, 50, 20, 4, -90) == po.ellipse2poly(50, 50, 20, 4, 270)
    assert po.ellipse2poly(7, 9, 0, 0, 33) == [(7, 9), (7, 9)]   # a single rounded point becomes the two-point polygon


def test_line_is_bresenham_left_to_right():
    assert po.line_pixels(50, 50, (2, 3), (7, 5)) == po.line_pixels(50, 50, (7, 5), (2, 3)) or \
        set(po.line_pixels(50, 50, (2, 3), (7, 5))) == set(po.line_pixels(50, 50, (7, 5), (2, 3)))
    px = po.line_pixels(50, 50, (2, 3), (7, 5))
    assert px[0] == (2, 3) and px[-1] == (7, 5) and len(px) == 6
    assert po.line_pixels(50, 50, (4, 4), (4, 4)) == [(4, 4)]
    v = po.line_pixels(50, 50, (10, 2), (8, 9))                  # steep: one pixel per row
    assert len(v) == 8 and sorted(y for _, y in v) == list(range(2, 10))
    # clipping keeps what is inside the image
    c = po.line_pixels(10, 10, (-5, 4), (4, 4))
    assert c == [(x, 4) for x in range(0, 5)]
    assert po.line_pixels(10, 10, (-5, -5), (-1, -9)) == []


def test_fill_covers_the_ellipse_and_every_row_is_one_run():
    rng = np.random.default_rng(0)
    H = W = 200
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(60):
        a, ang = int(rng.integers(0, 90)), int(rng.integers(-180, 181))
        cx, cy = int(rng.integers(95, 105)), int(rng.integers(95, 105))
        m = po.limb_mask(H, W, cx, cy, a, ang)
        t = np.radians(ang)
        u = (xx - cx) * np.cos(t) + (yy - cy) * np.sin(t)
        w = -(xx - cx) * np.sin(t) + (yy - cy) * np.cos(t)
        inner = (u / max(a - 1.0, 0.25)) ** 2 + (w / 3.0) ** 2 <= 1.0
        outer = (u / (a + 1.75)) ** 2 + (w / 5.75) ** 2 <= 1.0
        if a >= 2:
            assert not (inner & ~m).any(), (a, ang)              # everything well inside the ellipse is painted
        assert not (m & ~outer).any(), (a, ang)                  # nothing beyond the outline ring
        for row in m:                                            # contiguity: what the kernel's per-row spans rely on
            idx = np.flatnonzero(row)
            assert idx.size == 0 or idx[-1] - idx[0] + 1 == idx.size, (a, ang)


def test_limb_at_the_image_border_is_clipped_not_wrapped():
    m = po.limb_mask(64, 64, 2, 30, 20, 10)
    assert m.any() and m[:, 0].any() and not m[:, 40:].any()
    full = po.limb_mask(64, 164, 102, 30, 20, 10)[:, 100:]       # the same limb drawn with room to its left, cropped
    assert np.array_equal(m[:, :30], full[:, :30])
