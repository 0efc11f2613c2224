"""CPU restatement (numpy) of the matching stage in front of the hot path -- TEST INFRASTRUCTURE, not product code.

Only tests/ may import this file.  PARITY UNPINNED: the reference calls an external binary, DeepMatching 1.2.2
(/root/reference/para_gen.py:227-240, `./deepmatching img1 img2 -nt 0 -out f -ngh_rad 100`, fetched by
/root/reference/deepmatching/get_deepmatching.sh:3); neither its source nor any output of it is in the reference tree
and nothing may be fetched.  What is restated here is the PUBLISHED algorithm -- J. Revaud, P. Weinzaepfel,
Z. Harchaoui, C. Schmid, "DeepMatching: Hierarchical Deformable Dense Matching", IJCV 120(3), 2016 -- section by
section, with every free choice the paper leaves open written down below; the HIP implementation
(arap_flow_amd/csrc_dm/arapmatch.hip) mirrors THIS file and is tested against it.  The only anchors in the reference
are its call site and the output format it parses: lines `x1 y1 x2 y2 score index`, integers first
(/root/reference/para_gen.py:468-479).

Algorithm (paper section -> function):
  Sec. 3.1 "pixel descriptor"      descriptors(): gray, half resolution (the binary's default -downscale 1), Gaussian
                                    smoothing, gradient, 8 rectified orientation channels, smoothing, sigmoid, smoothing,
                                    ninth constant channel (0.3), L2 normalisation per pixel
  Sec. 3.2 "bottom-level maps"      level0(): every non-overlapping 4x4 patch of image 1 correlated with image 2 over the
                                    displacements |d| <= ngh_rad (the -ngh_rad option), mean of the 16 per-pixel dot products
  Sec. 3.2 Alg. 1 "pyramid"         level_up(): 3x3 max-pooling + subsampling by 2 of the children's maps, average of the
                                    four children, power rectification x^1.4
  Sec. 3.3 Alg. 2 "backtracking"    entries(), backtrack(): the maximum of every patch's map at every level above the
                                    bottom is an entry point; it is undone level by level (each child takes the best cell
                                    of its 3x3 pooling window, scores add up) down to atomic correspondences
  Sec. 3.3 "merging"                matches(): best candidate per atomic patch, then one match per 4x4 cell of image 2
                                    (the better one wins), coordinates back to full resolution

Coordinates.  A level-l map is indexed [ky][kx] over displacements (kx - c_l, ky - c_l) * 2^l pixels (half resolution),
c_0 = r = ngh_rad >> 1.  Subsampling keeps the displacement 0 on the grid: cell k of level l+1 is cell 2k + o_l of level l
with o_l = c_l mod 2, c_{l+1} = (c_l - o_l) / 2.
Patches.  Level 0: atomic 4x4 patches on the grid (w // 4) x (h // 4).  Level 1 (8x8): parent (I, J) = the atomic patches
(I, J), (I+1, J), (I, J+1), (I+1, J+1).  Level l+1 >= 2: parent (I, J) = the level-l patches (2I, 2J), (2I+2, 2J),
(2I, 2J+2), (2I+2, 2J+2) (level-l patches of level >= 1 sit on a grid of half their size, so these four tile the parent).
Only parents with all four children exist; levels are added while a level has at least one patch.
"""
import numpy as np

LAMBDA = np.float32(1.4)          # power rectification (paper Sec. 3.2)
NINTH = np.float32(0.3)           # constant ninth descriptor channel (Sec. 3.1)
SIGMOID = np.float32(0.2)         # slope of the sigmoid (Sec. 3.1)
PATCH = 4                         # atomic patch size
F32 = np.float32


def gauss7():
    """sigma = 1 Gaussian, radius 3, normalised (float32)"""
    x = np.arange(-3, 4, dtype=np.float64)
    g = np.exp(-x * x / 2.0)
    return (g / g.sum()).astype(F32)


def _blur(img):
    """separable 7-tap blur with replicated borders; img [h][w] or [h][w][c]; float32 accumulation in tap order -3..3,
    rows first then columns"""
    g = gauss7()
    h, w = img.shape[:2]
    xs = np.clip(np.arange(w)[None, :] + np.arange(-3, 4)[:, None], 0, w - 1)
    tmp = np.zeros_like(img, dtype=F32)
    for t in range(7):
        tmp = tmp + g[t] * img[:, xs[t]]
    ys = np.clip(np.arange(h)[None, :] + np.arange(-3, 4)[:, None], 0, h - 1)
    out = np.zeros_like(img, dtype=F32)
    for t in range(7):
        out = out + g[t] * tmp[ys[t]]
    return out.astype(F32)


def descriptors(rgb):
    """uint8 [H][W][3] -> float32 [H//2][W//2][9] unit-norm pixel descriptors (Sec. 3.1)"""
    H, W = rgb.shape[:2]
    h, w = H // 2, W // 2
    g = rgb[:2 * h, :2 * w].astype(F32)
    gray = (g[..., 0] + g[..., 1] + g[..., 2]) * F32(1.0 / 3.0)
    half = (gray[0::2, 0::2] + gray[0::2, 1::2] + gray[1::2, 0::2] + gray[1::2, 1::2]) * F32(0.25)
    sm = _blur(half)
    xp = np.clip(np.arange(w) + 1, 0, w - 1)
    xm = np.clip(np.arange(w) - 1, 0, w - 1)
    yp = np.clip(np.arange(h) + 1, 0, h - 1)
    ym = np.clip(np.arange(h) - 1, 0, h - 1)
    gx = (sm[:, xp] - sm[:, xm]) * F32(0.5)
    gy = (sm[yp] - sm[ym]) * F32(0.5)
    ang = np.arange(8, dtype=np.float64) * (np.pi / 4.0)
    cs, sn = np.cos(ang).astype(F32), np.sin(ang).astype(F32)
    ori = np.maximum(F32(0.0), gx[..., None] * cs + gy[..., None] * sn).astype(F32)
    ori = _blur(ori)
    ori = (F32(2.0) / (F32(1.0) + np.exp(-SIGMOID * ori, dtype=F32)) - F32(1.0)).astype(F32)
    ori = _blur(ori)
    d = np.concatenate([ori, np.full((h, w, 1), NINTH, F32)], axis=-1)
    n = np.sqrt((d * d).sum(-1, dtype=F32), dtype=F32)
    return (d / n[..., None]).astype(F32)


def level0(d1, d2, r):
    """bottom-level maps (Sec. 3.2): float32 [gh][gw][2r+1][2r+1]; patch (j, i) = pixels [4j, 4j+4) x [4i, 4i+4) of
    image 1; cell (dy + r, dx + r) = mean over the patch's 16 pixels of <d1(p), d2(p + (dx, dy))>, pixels of image 2
    outside the frame contributing 0"""
    h, w = d1.shape[:2]
    gh, gw = h // PATCH, w // PATCH
    S = 2 * r + 1
    out = np.zeros((gh, gw, S, S), F32)
    pad = np.zeros((h + 2 * r, w + 2 * r, 9), F32)
    pad[r:r + d2.shape[0], r:r + d2.shape[1]] = d2[:h, :w] if d2.shape[:2] != (h, w) else d2
    a = d1[:gh * PATCH, :gw * PATCH]
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            b = pad[r + dy:r + dy + gh * PATCH, r + dx:r + dx + gw * PATCH]
            dots = (a * b).sum(-1, dtype=F32)
            out[:, :, dy + r, dx + r] = dots.reshape(gh, PATCH, gw, PATCH).sum((1, 3), dtype=F32) * F32(1.0 / 16.0)
    return out


def pool_geometry(S, c):
    """subsampling of a level with map size S and centre cell c: (offset o, new size, new centre)"""
    o = c & 1
    return o, (S - 1 - o) // 2 + 1, (c - o) // 2


def maxpool(m, o):
    """[..., S, S] -> [..., S', S']: cell k = max over the 3x3 window around cell 2k + o (inside the map)"""
    S = m.shape[-1]
    S2 = (S - 1 - o) // 2 + 1
    neg = np.full(m.shape[:-2] + (S + 2, S + 2), -np.inf, F32)
    neg[..., 1:S + 1, 1:S + 1] = m
    out = np.full(m.shape[:-2] + (S2, S2), -np.inf, F32)
    for u in range(3):
        for v in range(3):
            out = np.maximum(out, neg[..., o + u:o + u + 2 * S2:2, o + v:o + v + 2 * S2:2])
    return out


def children_of(level, gh, gw):
    """patch grid of level `level` + 1 built on a level-`level` grid (gh, gw): (gh', gw', index arrays [4][gh'][gw'][2])"""
    if level == 0:
        nh, nw = gh - 1, gw - 1
        step = 1
        base = 1
    else:
        nh, nw = ((gh - 3) // 2 + 1 if gh >= 3 else 0), ((gw - 3) // 2 + 1 if gw >= 3 else 0)
        step = 2
        base = 2
    if nh <= 0 or nw <= 0:
        return 0, 0, None
    J, I = np.mgrid[0:nh, 0:nw]
    kids = []
    for b in range(2):
        for a in range(2):
            kids.append(np.stack([J * base + b * step, I * base + a * step], -1))
    return nh, nw, np.stack(kids)                              # order: (0,0), (0,+x), (+y,0), (+y,+x)


def level_up(m, level, c):
    """one iteration of Alg. 1: maps [gh][gw][S][S] of level `level` -> (maps of level + 1, kids, o, c')"""
    gh, gw, S, _ = m.shape
    nh, nw, kids = children_of(level, gh, gw)
    if kids is None:
        return None
    o, S2, c2 = pool_geometry(S, c)
    pooled = maxpool(m, o)
    acc = np.zeros((nh, nw, S2, S2), F32)
    for q in range(4):
        acc = acc + pooled[kids[q][..., 0], kids[q][..., 1]]
    acc = acc * F32(0.25)
    return np.power(np.maximum(acc, F32(0.0)), LAMBDA, dtype=F32).astype(F32), kids, o, c2


def pyramid(d1, d2, r):
    """list of levels: dict(maps, c, kids (to the level below), o (subsampling offset of the level below))"""
    levels = [dict(maps=level0(d1, d2, r), c=r, kids=None, o=None)]
    while True:
        cur = levels[-1]
        up = level_up(cur["maps"], len(levels) - 1, cur["c"])
        if up is None or up[0].shape[-1] < 1:
            break
        maps, kids, o, c2 = up
        levels.append(dict(maps=maps, c=c2, kids=kids, o=o))
        if maps.shape[-1] == 1:
            break
    return levels


def backtrack(levels):
    """Alg. 2: every patch of every level >= 1 enters with the first maximum of its map; undone down to level 0.
    Returns per atomic patch the best candidate: score [gh][gw] (0 = none) and cell (ky, kx) [gh][gw][2]."""
    gh, gw, S0, _ = levels[0]["maps"].shape
    best = np.zeros((gh, gw), F32)
    cell = np.zeros((gh, gw, 2), np.int32)
    code = np.full((gh, gw), np.iinfo(np.int64).max, np.int64)
    for top in range(1, len(levels)):
        m = levels[top]["maps"]
        nh, nw, S, _ = m.shape
        flat = m.reshape(nh, nw, S * S)
        k = flat.argmax(-1)
        # work list of the level: (j, i, ky, kx, score)
        J, I = np.mgrid[0:nh, 0:nw]
        cur = np.stack([J.ravel(), I.ravel(), (k // S).ravel(), (k % S).ravel()], -1).astype(np.int64)
        sc = np.take_along_axis(flat, k[..., None], -1)[..., 0].ravel().astype(F32)
        for lv in range(top, 0, -1):
            kids, o = levels[lv]["kids"], levels[lv]["o"]
            below = levels[lv - 1]["maps"]
            Sb = below.shape[-1]
            nxt, nsc = [], []
            for q in range(4):
                cj = kids[q][cur[:, 0], cur[:, 1], 0]
                ci = kids[q][cur[:, 0], cur[:, 1], 1]
                bv = np.full(len(cur), -np.inf, F32)
                by = np.zeros(len(cur), np.int64)
                bx = np.zeros(len(cur), np.int64)
                for u in (-1, 0, 1):
                    for v in (-1, 0, 1):
                        y = 2 * cur[:, 2] + o + u
                        x = 2 * cur[:, 3] + o + v
                        ok = (y >= 0) & (y < Sb) & (x >= 0) & (x < Sb)
                        val = np.where(ok, below[cj, ci, np.clip(y, 0, Sb - 1), np.clip(x, 0, Sb - 1)], -np.inf).astype(F32)
                        better = val > bv                             # first maximum in (u, v) order
                        bv = np.where(better, val, bv)
                        by = np.where(better, y, by)
                        bx = np.where(better, x, bx)
                nxt.append(np.stack([cj, ci, by, bx], -1))
                nsc.append((sc + bv).astype(F32))
            cur = np.concatenate(nxt)
            sc = np.concatenate(nsc)
        # level 0 candidates: keep the best per atomic patch (ties: the smaller cell index)
        for (j, i, ky, kx), s in zip(cur, sc):
            cd = ky * S0 + kx
            if s > best[j, i] or (s == best[j, i] and s > 0 and cd < code[j, i]):
                best[j, i] = s
                cell[j, i] = (ky, kx)
                code[j, i] = cd
    return best, cell


def matches(rgb1, rgb2, ngh_rad=100):
    """the whole stage: float array [n][6] = x1 y1 x2 y2 score index (full-resolution pixel coordinates, the order of
    the atomic patches), as the binary prints them"""
    r = int(ngh_rad) >> 1
    d1, d2 = descriptors(rgb1), descriptors(rgb2)
    h, w = d1.shape[:2]
    levels = pyramid(d1, d2, r)
    best, cell = backtrack(levels)
    gh, gw = best.shape
    c0 = levels[0]["c"]
    # one match per 4x4 cell of image 2: the better score wins, ties -> the smaller atomic patch index
    h2, w2 = d2.shape[:2]
    win = {}
    for j in range(gh):
        for i in range(gw):
            if best[j, i] <= 0:
                continue
            x2 = PATCH * i + 2 + int(cell[j, i, 1]) - c0
            y2 = PATCH * j + 2 + int(cell[j, i, 0]) - c0
            if not (0 <= x2 < w2 and 0 <= y2 < h2):
                continue
            key = (y2 // PATCH, x2 // PATCH)
            cand = (float(best[j, i]), -(j * gw + i))
            if key not in win or cand > win[key][0]:
                win[key] = (cand, (j, i, x2, y2))
    out = []
    for (_, (j, i, x2, y2)) in sorted(win.values(), key=lambda t: -t[0][1]):
        out.append((2 * (PATCH * i + 2), 2 * (PATCH * j + 2), 2 * x2, 2 * y2, float(best[j, i]), len(out)))
    return np.asarray(out, np.float32).reshape(-1, 6)
