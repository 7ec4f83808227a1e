"""CPU restatement (numpy + plain loops) of the OpenPose control-map drawing of include/gip_pose.h — TEST INFRASTRUCTURE ONLY.

Follows threestudio/utils/poser.py:832-904 (Skeleton.openpose_draw): discs in key-point order (cv2.circle, radius 4,
filled), then limbs in line order, each `cv2.ellipse2Poly((int(mX), int(mY)), (int(length / 2), 4), int(angle), 0, 360, 1)`
+ `cv2.fillConvexPoly` on a copy, blended 0.4 / 0.6 into the canvas with `cv2.addWeighted` (uint8, round half to even).

The drawing routines live in a third-party dependency that is absent from /root/reference AND not installed here:
opencv-python (requirements.txt pins `opencv-python`, no version).  They are restated below from OpenCV 4.x's published
source, modules/imgproc/src/drawing.cpp, function by function:
  * SinTable / sincos / ellipse2Poly (double overload, then the integer overload's cvRound + duplicate removal);
  * Circle (filled, the midpoint variant used for thickness < 0, shift 0);
  * clipLine, LineIterator (8-connected, left_to_right) = Line;
  * FillConvexPoly (outline by Line, then the XY_SHIFT = 16 fixed-point scanline fill).
PARITY AGAINST THE OPENCV BINARY IS UNPINNED (it cannot run here); what is pinned: the HIP kernel bit-exactly against
THIS file, and this file against hand-checkable properties (tests/test_pose_oracle.py: the radius-4 disc rows, polygon
vertex counts / symmetry, fill == brute-force point-in-polygon up to the outline ring, degenerate and clipped limbs).
Only tests/ may import this module; the product path never does."""
import math

import numpy as np

COLORS = np.array([[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0],
                   [0, 255, 85], [0, 255, 170], [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255],
                   [170, 0, 255], [255, 0, 255], [255, 0, 170], [255, 0, 85]], np.uint8)      # poser.py:701-703
LINES = np.array([[0, 1], [1, 2], [2, 3], [3, 4], [1, 5], [5, 6], [6, 7], [1, 8], [8, 9], [9, 10], [1, 11], [11, 12],
                  [12, 13], [0, 14], [14, 16], [0, 15], [15, 17]], np.int32)                    # poser.py:688
XY_SHIFT = 16
XY_ONE = 1 << XY_SHIFT

# drawing.cpp: `static const float SinTable[]`, 451 literals = sin(degrees) printed with seven decimals
SIN_TABLE = np.array([round(math.sin(math.radians(a)), 7) for a in range(451)], dtype=np.float64).astype(np.float32)


def cv_round(x):
    """cvRound(double): round half to even (lrint under the default rounding mode)."""
    return int(np.rint(np.float64(x)))


def ellipse2poly(cx, cy, a, b, angle, arc_start=0, arc_end=360, delta=1):
    """cv::ellipse2Poly(Point center, Size axes, int angle, int arcStart, int arcEnd, int delta, vector<Point>&)."""
    while angle < 0:
        angle += 360
    while angle > 360:
        angle -= 360
    if arc_start > arc_end:
        arc_start, arc_end = arc_end, arc_start
    while arc_start < 0:
        arc_start += 360
        arc_end += 360
    while arc_end > 360:
        arc_end -= 360
        arc_start -= 360
    if arc_end - arc_start > 360:
        arc_start, arc_end = 0, 360
    alpha, beta = np.float64(SIN_TABLE[450 - angle]), np.float64(SIN_TABLE[angle])      # sincos(angle, alpha = cos, beta = sin): float table
    pts_d = []
    i = arc_start
    while i < arc_end + delta:
        ang = min(i, arc_end)
        if ang < 0:
            ang += 360
        x = np.float64(a) * np.float64(SIN_TABLE[450 - ang])
        y = np.float64(b) * np.float64(SIN_TABLE[ang])
        pts_d.append((np.float64(cx) + x * alpha - y * beta, np.float64(cy) + x * beta + y * alpha))
        i += delta
    pts, prev = [], None
    for px, py in pts_d:                       # the integer overload: cvRound, consecutive duplicates dropped
        p = (cv_round(px), cv_round(py))
        if p != prev:
            pts.append(p)
            prev = p
    if len(pts) == 1:
        pts = [(int(cx), int(cy)), (int(cx), int(cy))]
    return pts


def clip_line(W, H, p1, p2):
    """cv::clipLine(Size2l, Point2l&, Point2l&) -> (inside?, p1, p2)."""
    x1, y1, x2, y2 = p1[0], p1[1], p2[0], p2[1]
    right, bottom = W - 1, H - 1
    if W <= 0 or H <= 0:
        return False, p1, p2
    c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8
    c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))          # (int64)((double)(a - y1) * (x2 - x1) / (y2 - y1)): truncation
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (c1 | c2) == 0, (x1, y1), (x2, y2)


def line_pixels(W, H, p1, p2):
    """Line(img, pt1, pt2, color, 8): clipLine + LineIterator(connectivity 8, left_to_right = true)."""
    ok, p1, p2 = clip_line(W, H, p1, p2)
    if not ok:
        return []
    dx, dy = p2[0] - p1[0], p2[1] - p1[1]
    sx, sy = 1, 1
    x, y = p1
    if dx < 0:                      # left_to_right: start from the right-hand end's mirror, i.e. swap the ends
        dx, dy = -dx, -dy
        x, y = p2
    if dy < 0:
        dy, sy = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
    err = dx - (dy + dy)
    plus_delta, minus_delta = dx + dx, -(dy + dy)
    out = []
    for _ in range(dx + 1):
        out.append((x, y))
        mask = err < 0
        err += minus_delta + (plus_delta if mask else 0)
        if vert:                    # the major axis is y
            y += sy
            if mask:
                x += sx
        else:
            x += sx
            if mask:
                y += sy
    return out


def _trunc_div(a, b):
    """C integer division (toward zero) on int64 values."""
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def fill_convex_poly_mask(H, W, pts):
    """FillConvexPoly(img, v, npts, color, LINE_8, shift 0) as the boolean footprint it paints."""
    mask = np.zeros((H, W), bool)
    n = len(pts)
    p0 = pts[n - 1]
    xmin = xmax = pts[0][0]
    ymin = ymax = pts[0][1]
    imin = 0
    for i, p in enumerate(pts):
        if p[1] < ymin:
            ymin, imin = p[1], i
        ymax, xmax, xmin = max(ymax, p[1]), max(xmax, p[0]), min(xmin, p[0])
        for (px, py) in line_pixels(W, H, p0, p):
            mask[py, px] = True
        p0 = p
    if n < 3 or xmax < 0 or ymax < 0 or xmin >= W or ymin >= H:
        return mask
    ymax = min(ymax, H - 1)
    edge = [dict(idx=imin, di=1, x=-XY_ONE, dx=0, ye=ymin), dict(idx=imin, di=n - 1, x=-XY_ONE, dx=0, ye=ymin)]
    edges = n
    y = ymin
    delta1 = delta2 = XY_ONE >> 1
    while True:
        for e in edge:
            if y >= e["ye"]:
                idx0, di = e["idx"], e["di"]
                idx = idx0 + di
                if idx >= n:
                    idx -= n
                while True:
                    edges -= 1                              # `for (; edges-- > 0; )`
                    if edges + 1 <= 0:
                        break
                    ty = pts[idx][1]
                    if ty > y:
                        xs, xe = pts[idx0][0] << XY_SHIFT, pts[idx][0] << XY_SHIFT
                        e["ye"] = ty
                        e["dx"] = _trunc_div((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e["x"] = xs
                        e["idx"] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= n:
                        idx -= n
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if edge[0]["x"] > edge[1]["x"] else (0, 1)
            xx1 = (edge[left]["x"] + delta1) >> XY_SHIFT
            xx2 = (edge[right]["x"] + delta2) >> XY_SHIFT
            if xx2 >= 0 and xx1 < W:
                xx1, xx2 = max(xx1, 0), min(xx2, W - 1)
                if xx1 <= xx2:
                    mask[y, xx1:xx2 + 1] = True
        edge[0]["x"] += edge[0]["dx"]
        edge[1]["x"] += edge[1]["dx"]
        y += 1
        if y > ymax:
            break
    return mask


def disc_mask(H, W, cx, cy, radius=4):
    """Circle(img, center, radius, color, fill = true): the midpoint variant of cv::circle for thickness < 0."""
    mask = np.zeros((H, W), bool)

    def hline(y, x1, x2):
        if 0 <= y < H:
            x1, x2 = max(x1, 0), min(x2, W - 1)
            if x1 <= x2:
                mask[y, x1:x2 + 1] = True
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        hline(cy - dy, cx - dx, cx + dx)
        hline(cy + dy, cx - dx, cx + dx)
        hline(cy - dx, cx - dy, cx + dy)
        hline(cy + dx, cx - dy, cx + dy)
        dy += 1
        err += plus
        plus += 2
        m = -1 if err > 0 else 0                 # mask = (err <= 0) - 1
        err -= minus & m
        dx += m
        minus -= m & 2
    return mask


def limb_mask(H, W, cx, cy, a, angle):
    return fill_convex_poly_mask(H, W, ellipse2poly(int(cx), int(cy), int(a), 4, int(angle)))


def draw(points_px, visible, limbs, H, W):
    """points_px [18,2] int, visible [18] bool, limbs [17,6] float32 (int centre x, int centre y, int(len / 2), drawn?,
    int angle in degrees, unused) -> [H,W,3] float32."""
    canvas = np.zeros((H, W, 3), np.uint8)
    for i in range(18):                                                           # poser.py:879-882
        if visible[i]:
            canvas[disc_mask(H, W, int(points_px[i, 0]), int(points_px[i, 1]))] = COLORS[i]
    for l in range(17):                                                           # poser.py:885-899
        cx, cy, a, on, ang = [float(v) for v in limbs[l][:5]]
        if on == 0:
            continue
        inside = limb_mask(H, W, cx, cy, a, ang)
        src = np.where(inside[..., None], COLORS[l][None, None, :], canvas)
        t = canvas.astype(np.float32) * np.float32(0.4) + src.astype(np.float32) * np.float32(0.6)
        canvas = np.clip(np.rint(t), 0, 255).astype(np.uint8)                    # cv2.addWeighted: saturate_cast<uchar>(round half to even)
    return canvas.astype(np.float32) / np.float32(255.0)
