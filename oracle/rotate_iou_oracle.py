"""CPU ORACLE for the evaluator-side overlap kernels (test infrastructure only; same rules as dd3d_oracle.py).

Restates tridet/evaluators/rotate_iou.py: rbbox_to_corners (:214-236), point_in_quadrilateral (:161-180),
line_segment_intersection (:81-124), quadrilateral_intersection (:183-211), sort_vertex_in_convex_polygon (:38-78), area (:30-35),
inter / devRotateIoUEval (:239-258), rotate_iou_gpu_eval (:292-327), d3_box_overlap_kernel (:330-357), image_box_overlap (:360-381).
All arithmetic in float32, in the reference's order.  Pinned by tests/golden/rotate_iou.npz, which tests/golden/
make_rotate_iou_golden.py produced by executing the reference's own functions.
"""
import math

import numpy as np

F = np.float32


def _corners(box):
    """:214-236 -- clockwise corners (-x,-y), (-x,+y), (+x,+y), (+x,-y) halves, rotated by +angle clockwise."""
    cx, cy, xd, yd, ang = (F(v) for v in box)
    c, s = F(math.cos(ang)), F(math.sin(ang))
    hx, hy = xd / F(2), yd / F(2)
    pts = []
    for sx, sy in ((-1, -1), (-1, 1), (1, 1), (1, -1)):
        px, py = F(sx) * hx, F(sy) * hy
        pts.append((c * px + s * py + cx, -s * px + c * py + cy))
    return pts


def _inside(p, quad):
    """:161-180 -- projections of AP on AB and AD lie inside [0, |AB|^2] x [0, |AD|^2] (eps 1e-4)."""
    a, b, d = quad[0], quad[1], quad[3]
    ab = (b[0] - a[0], b[1] - a[1])
    ad = (d[0] - a[0], d[1] - a[1])
    ap = (p[0] - a[0], p[1] - a[1])
    abab, abap = ab[0] * ab[0] + ab[1] * ab[1], ab[0] * ap[0] + ab[1] * ap[1]
    adad, adap = ad[0] * ad[0] + ad[1] * ad[1], ad[0] * ap[0] + ad[1] * ap[1]
    eps = F(0.0001)
    return abab >= abap - eps and abap >= F(0) - eps and adad >= adap - eps and adap >= F(0) - eps


def _segment_cross(A, B, C, D):
    """:81-124 -- orientation tests, then the crossing point by Cramer's rule; None when the segments do not cross."""
    BA0, BA1 = B[0] - A[0], B[1] - A[1]
    DA0, DA1 = D[0] - A[0], D[1] - A[1]
    CA0, CA1 = C[0] - A[0], C[1] - A[1]
    acd = DA1 * CA0 > CA1 * DA0
    bcd = (D[1] - B[1]) * (C[0] - B[0]) > (C[1] - B[1]) * (D[0] - B[0])
    if acd == bcd:
        return None
    abc = CA1 * BA0 > BA1 * CA0
    abd = DA1 * BA0 > BA1 * DA0
    if abc == abd:
        return None
    DC0, DC1 = D[0] - C[0], D[1] - C[1]
    ABBA = A[0] * B[1] - B[0] * A[1]
    CDDC = C[0] * D[1] - D[0] * C[1]
    DH = BA1 * DC0 - BA0 * DC1
    return ((ABBA * DC0 - BA0 * CDDC) / DH, (ABBA * DC1 - BA1 * CDDC) / DH)


def intersection_area(box1, box2):
    """inter(rbbox1, rbbox2) :239-251."""
    q1, q2 = _corners(box1), _corners(box2)
    pts = []
    for i in range(4):  # :185-194, vertices of either inside the other, interleaved
        if _inside(q1[i], q2):
            pts.append(q1[i])
        if _inside(q2[i], q1):
            pts.append(q2[i])
    for i in range(4):  # :196-203
        for j in range(4):
            x = _segment_cross(q1[i], q1[(i + 1) % 4], q2[j], q2[(j + 1) % 4])
            if x is not None:
                pts.append(x)
    n = len(pts)
    if n == 0:
        return F(0)
    # :38-78 -- order by a monotone pseudo-angle of (p - centroid): key = x/|v| for y >= 0, -2 - x/|v| below; insertion sort
    cx = F(0)
    cy = F(0)
    for p in pts:
        cx += p[0]
        cy += p[1]
    cx /= F(n)
    cy /= F(n)
    keys = []
    for p in pts:
        vx, vy = p[0] - cx, p[1] - cy
        d = F(math.sqrt(vx * vx + vy * vy))
        vx, vy = vx / d, vy / d
        keys.append(F(-2) - vx if vy < 0 else vx)
    pts = [list(p) for p in pts]
    for i in range(1, n):
        if keys[i - 1] > keys[i]:
            k, p = keys[i], pts[i]
            j = i
            while j > 0 and keys[j - 1] > k:
                keys[j], pts[j] = keys[j - 1], pts[j - 1]
                j -= 1
            keys[j], pts[j] = k, p
    # :30-35 -- triangle fan from the first vertex
    total = F(0)
    a = pts[0]
    for i in range(n - 2):
        b, c = pts[i + 1], pts[i + 2]
        total += abs(((a[0] - c[0]) * (b[1] - c[1]) - (a[1] - c[1]) * (b[0] - c[0])) / F(2))
    return F(total)


def rotate_iou_eval(boxes, query_boxes, criterion=-1):
    """rotate_iou_gpu_eval :292-327: out[i, j] = devRotateIoUEval(query j, box i) (:289): criterion -1 IoU, 0 inter / area(query),
    1 inter / area(box), else the intersection area."""
    boxes, query_boxes = np.asarray(boxes, dtype=F), np.asarray(query_boxes, dtype=F)
    out = np.zeros((len(boxes), len(query_boxes)), dtype=F)
    for i, b in enumerate(boxes):
        for j, q in enumerate(query_boxes):
            a1, a2 = q[2] * q[3], b[2] * b[3]  # rbox1 = query, rbox2 = box
            it = intersection_area(q, b)
            if criterion == -1:
                out[i, j] = it / (a1 + a2 - it)
            elif criterion == 0:
                out[i, j] = it / a1
            elif criterion == 1:
                out[i, j] = it / a2
            else:
                out[i, j] = it
    return out


def d3_box_overlap(boxes, qboxes, rinc, criterion=-1, camera_coordinate=False):
    """d3_box_overlap_kernel :330-357 (returns a new array): rinc = BEV intersection areas; multiplied by the vertical overlap."""
    boxes, qboxes = np.asarray(boxes, dtype=F), np.asarray(qboxes, dtype=F)
    out = np.array(rinc, dtype=F, copy=True)
    for i in range(len(boxes)):
        for j in range(len(qboxes)):
            if out[i, j] > 0:
                b, q = boxes[i], qboxes[j]
                if camera_coordinate:
                    iw = min(b[1], q[1]) - max(b[1] - b[4], q[1] - q[4])
                else:
                    iw = min(b[2] + b[5], q[2] + q[5]) - max(b[2], q[2])
                if iw > 0:
                    v1, v2 = b[3] * b[4] * b[5], q[3] * q[4] * q[5]
                    inc = iw * out[i, j]
                    ua = (v1 + v2 - inc) if criterion == -1 else v1 if criterion == 0 else v2 if criterion == 1 else inc
                    out[i, j] = inc / ua
                else:
                    out[i, j] = 0.0
    return out


def image_box_overlap(boxes, query_boxes, criterion=-1):
    """:360-381."""
    boxes, query_boxes = np.asarray(boxes, dtype=F), np.asarray(query_boxes, dtype=F)
    out = np.zeros((len(boxes), len(query_boxes)), dtype=F)
    for k, q in enumerate(query_boxes):
        qa = (q[2] - q[0]) * (q[3] - q[1])
        for n, b in enumerate(boxes):
            iw = min(b[2], q[2]) - max(b[0], q[0])
            ih = min(b[3], q[3]) - max(b[1], q[1])
            if iw > 0 and ih > 0:
                ba = (b[2] - b[0]) * (b[3] - b[1])
                ua = (ba + qa - iw * ih) if criterion == -1 else ba if criterion == 0 else qa if criterion == 1 else F(1)
                out[n, k] = iw * ih / ua
    return out
