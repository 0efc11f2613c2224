"""Seeded synthetic DAVIS-shaped inputs for the bench and the parity tests.

What para_gen.py would hand to arap_deform for one frame pair: an RGB frame, an inverted object mask
(0 on the object, 255 elsewhere: para_gen.py:514-528) and the sparse matches that survive its filter
(same segment, 0 < |d| < 60: para_gen.py:216-223) as `x1 y1 x2 y2` rows (para_gen.py:476-479).
DeepMatching itself is not in the reference tree, so the matches are synthesised: a lattice of handles
every `step` px inside each segment displaced by a per-segment similarity transform plus jitter.
"""
import numpy as np


def _ellipse_mask(W, H, cx, cy, ax, ay, theta):
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    c, s = np.cos(theta), np.sin(theta)
    u = (xs - cx) * c + (ys - cy) * s
    v = -(xs - cx) * s + (ys - cy) * c
    return (u / ax) ** 2 + (v / ay) ** 2 <= 1.0


def make_labels(W, H, K, seed, area_frac=0.25):
    """K disjoint ellipses, total area ~ area_frac of the frame, none touching the border.
    Returns int32 label image, 0 = background, 1..K = segments."""
    rng = np.random.default_rng(seed)
    labels = np.zeros((H, W), np.int32)
    per = area_frac / K
    for k in range(1, K + 1):
        for _ in range(200):
            ratio = rng.uniform(0.6, 1.6)
            ay = np.sqrt(per * W * H / (np.pi * ratio))
            ax = ratio * ay
            m = max(ax, ay) + 3
            if 2 * m >= min(W, H):
                ax *= 0.5; ay *= 0.5; m = max(ax, ay) + 3
            cx, cy = rng.uniform(m, W - m), rng.uniform(m, H - m)
            e = _ellipse_mask(W, H, cx, cy, ax, ay, rng.uniform(0, np.pi))
            if not (labels[e] != 0).any():
                labels[e] = k
                break
    return labels


def make_constraints(labels, seed, fd=1, step=8):
    """Lattice handles inside each segment, similarity displacement per segment + jitter, rounded to
    ints; dropped if |d| >= 60, |d| == 0 or the target leaves the frame (para_gen.py:216-223)."""
    rng = np.random.default_rng(seed + 7919)
    H, W = labels.shape
    rows = []
    for k in range(1, int(labels.max()) + 1):
        ys, xs = np.nonzero(labels == k)
        if len(xs) == 0:
            continue
        cx, cy = xs.mean(), ys.mean()
        rot = rng.normal(0, np.deg2rad(2.0 * fd))
        sc = rng.normal(1.0, 0.01 * fd)
        t = rng.normal(0, 3.0 * fd, 2)
        for y in range(step // 2, H, step):
            for x in range(step // 2, W, step):
                if labels[y, x] != k:
                    continue
                dx, dy = x - cx, y - cy
                tx = cx + sc * (np.cos(rot) * dx - np.sin(rot) * dy) + t[0] + rng.normal(0, 0.5)
                ty = cy + sc * (np.sin(rot) * dx + np.cos(rot) * dy) + t[1] + rng.normal(0, 0.5)
                tx, ty = int(round(tx)), int(round(ty))
                d2 = (tx - x) ** 2 + (ty - y) ** 2
                if d2 == 0 or d2 >= 3600 or not (0 <= tx < W and 0 <= ty < H):
                    continue
                rows.append((x, y, tx, ty))
    return np.asarray(rows, np.int32).reshape(-1, 4)


def make_rgb(W, H, seed):
    rng = np.random.default_rng(seed + 104729)
    img = rng.integers(0, 256, (H // 8 + 2, W // 8 + 2, 3)).astype(np.float32)
    img = np.kron(img, np.ones((8, 8, 1), np.float32))[:H, :W]
    img = (img + np.roll(img, 3, 0) + np.roll(img, 3, 1) + np.roll(img, (2, 2), (0, 1))) / 4.0
    return img.astype(np.uint8)


def make_frame(W=854, H=480, seed=0, K=1, fd=1, full_mask=False):
    """One synthetic frame.  Returns dict(rgb u8[H,W,3], mask_red u8[H,W], constraints int32[n,4],
    labels).  full_mask=True is the roofline configuration: mask == 0 everywhere (all vertices
    active), constraints on a lattice over the whole frame."""
    if full_mask:
        labels = np.ones((H, W), np.int32)
    else:
        labels = make_labels(W, H, K, seed)
    cons = make_constraints(labels, seed, fd=fd)
    mask_red = np.where(labels != 0, 0, 255).astype(np.uint8)
    return dict(rgb=make_rgb(W, H, seed), mask_red=mask_red, constraints=cons, labels=labels)


def segment_masks(frame):
    """--multseg split (para_gen.py:518-540): one inverted mask and one constraint list per label."""
    out = []
    labels, cons = frame["labels"], frame["constraints"]
    for k in range(1, int(labels.max()) + 1):
        m = np.where(labels == k, 0, 255).astype(np.uint8)
        sel = labels[cons[:, 1], cons[:, 0]] == k if len(cons) else np.zeros(0, bool)
        out.append(dict(rgb=frame["rgb"], mask_red=m, constraints=cons[sel], label=k))
    return out
