"""Python 3 host side of the dataset generator, over libarapopt.so.

What the reference's Python 2 scripts and C++ drivers do AROUND the hot path, written from their
behaviour (the contract each function states in its docstring), with the size / selection arithmetic as pure
functions that tests/golden/host/ pins with hand-computed cases:
  read_list / deform_list   ARAP/deformation/src/main.cpp:162-241  (arap_deform: 6 paths per line)
  warp_files                ARAP/warping/src/main.cpp:302-336      (warp_image)
  cover_scale, fit_bg, add_bg            para_gen.py:36-61      (background compositing)
  merge_segments, flatten                para_gen.py:136-175    (--multseg: merge per-segment outputs by the warped masks)
  match_ok, valid_cnstr, filter_matches  para_gen.py:216-223,468-482
  resize_crop_geometry, scale_rotate     para_gen.py:253-291
  make_arap_path                         para_gen.py:331-339
No oracle import; the solve and the rasteriser run on the GPU through arap_flow_amd.opt.
"""
import os
import os.path as osp
import random as rn

import numpy as np
from PIL import Image

from . import flo

ARAP_BG = 255        # para_gen.py:30

_ANTIALIAS = getattr(Image, "LANCZOS", None) or Image.ANTIALIAS   # Image.ANTIALIAS of the reference's PIL


# ------------------------------------------------------------------------------------------------------
# arap_deform
# ------------------------------------------------------------------------------------------------------
def read_list(path):
    """main.cpp:183-191: one solve per line, six whitespace-separated paths
    rgb mask constraints out_flow out_rgb out_mask"""
    lines = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if len(tok) < 6:
                raise ValueError("list line needs 6 paths: %r" % line)
            lines.append(tuple(tok[:6]))
    return lines


def load_rgb(path):
    return np.array(Image.open(path).convert("RGB"))


def load_mask_red(path):
    """red channel of the mask PNG (CombinedSolver.h:213,234): 0 = deformable object"""
    return np.array(Image.open(path).convert("RGB"))[..., 0]


def save_mask(mask, path):
    """LodePNG writes the 0/255 warped mask as a 1-bit image: np.array(Image.open()) of
    it is bool, which para_gen.py's flatten relies on only through != 0 / == 0."""
    Image.fromarray(np.ascontiguousarray(mask) > 0).save(path)


FILL_MIN, FILL_MAX = 8, 32     # frames per solve call: at least / at most (see deform_list)


def _load_line(ln):
    """loadData (main.cpp:116-138) of one list line: RGB, red channel of the mask, constraint rows"""
    from . import opt
    return load_rgb(ln[0]), load_mask_red(ln[1]), opt.load_constraints(ln[2])


def _save_result(ln, r):
    Image.fromarray(r["warped_rgb"]).save(ln[4])
    save_mask(r["warped_mask"], ln[5])
    flo.flow_write(ln[3], r["flow"])


def deform_list(state, lines, num_iter=19, non_linear_iter=8, linear_iter=400, max_batch=FILL_MAX, verbose=True):
    """arap_deform over a list (main.cpp:223-238).  Frames of equal size are solved together: the library gives
    every solve a group of the resident launch's workgroups sized by its active tiles, and a launch costs the same
    however full it is, so frames are added to a batch while they still fit ONE launch (8 DAVIS-shaped 854x480
    frames, ~24 --multseg segment solves); FILL_MIN frames per call when the resident path does not apply.
    The GPU does not wait for the host (the structure of arap_flow_amd/host/arap_deform.cpp): two solver objects
    alternate; while one batch is solved the next is decoded (worker threads) and uploaded into the other object, and
    the previous batch's results are read from pinned memory and encoded (worker threads)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import opt
    state.use_own_stream()
    ahead = 2 * max_batch
    with ThreadPoolExecutor(max_workers=8) as pool:
        loading = {}

        def frame_at(k):
            for q in range(k, min(len(lines), k + ahead + 1)):
                if q not in loading:
                    loading[q] = pool.submit(_load_line, lines[q])
            return loading[k].result()

        writing = []
        lanes = [dict(solver=None, batch=[]), dict(solver=None, batch=[])]
        size = None

        def drain(lane):
            """wait for the lane's solve, hand copies of its results to the writer threads"""
            if not lane["batch"]:
                return
            lane["solver"].wait()
            for b, ln in enumerate(lane["batch"]):
                r = lane["solver"].host_results(b)
                res = dict(flow=r["flow"].copy(), warped_rgb=r["warped_rgb"].copy(), warped_mask=r["warped_mask"].copy())
                writing.append(pool.submit(_save_result, ln, res))
                if verbose:
                    print("Saved")                                          # main.cpp:159
            lane["batch"] = []
            while len(writing) > 48:
                writing.pop(0).result()

        i, cur = 0, 0
        while i < len(lines):
            rgb0 = frame_at(i)[0]
            H, W = rgb0.shape[:2]
            if size != (W, H):
                for lane in lanes:
                    drain(lane)
                    if lane["solver"] is not None:
                        lane["solver"].close()
                if size is not None and verbose:
                    print("Warning: Input image has different size to one in the prebuilt plan.\n"
                          "To avoid re-building the plan and to save time, put images of the same size in the "
                          "same list.\nStarting to re-build plan...")      # CombinedSolver.h:151-153
                for lane in lanes:
                    lane["solver"] = opt.FrameSolver(state, W, H, batch=max_batch)
                size = (W, H)
            lane, other = lanes[cur], lanes[cur ^ 1]
            solver = lane["solver"]
            batch = []
            j = i
            while j < len(lines) and len(batch) < max_batch:
                ln = lines[j]
                rgb, mask, cons = frame_at(j)
                if rgb.shape[:2] != (H, W):
                    break
                if mask.shape != (H, W):
                    raise ValueError("mask %s has another size than %s" % (ln[1], ln[0]))
                b = len(batch)
                solver.set_frame(b, mask, cons, rgb=rgb, border_pins=True)
                if b > 0:
                    launches = solver.launches_for(b + 1)
                    if launches > 1 or (launches == 0 and b >= FILL_MIN):
                        break                                   # this frame opens the next batch (its slot is re-set)
                batch.append(ln)
                del loading[j]                                  # the device holds it now
                j += 1
            solver.solve_async(len(batch), num_iter, non_linear_iter, linear_iter, warp=True, download=True)
            lane["batch"] = batch
            drain(other)                                        # the previous batch, while this one is being solved
            cur ^= 1
            i = j
        for lane in (lanes[cur], lanes[cur ^ 1]):
            drain(lane)
        for f in writing:
            f.result()
        for lane in lanes:
            if lane["solver"] is not None:
                lane["solver"].close()


def warp_files(state, rgb_path, mask_path, flo_path, out_rgb_path, out_mask_path):
    """warp_image (ARAP/warping/src/main.cpp:302-336)"""
    from . import opt
    rgb, mask, fl = load_rgb(rgb_path), load_mask_red(mask_path), flo.flow_read(flo_path)
    if fl.shape[:2] != mask.shape or rgb.shape[:2] != mask.shape:
        raise ValueError("image, mask and flow sizes differ")
    wrgb, wmsk = opt.warp_image(state, rgb, mask, fl)
    Image.fromarray(wrgb).save(out_rgb_path)
    save_mask(wmsk, out_mask_path)


# ------------------------------------------------------------------------------------------------------
# para_gen helpers
# ------------------------------------------------------------------------------------------------------
def cover_scale(bg_hw, im_hw, u):
    """Background compositing, size arithmetic (behaviour of /root/reference/para_gen.py:36-48): the background is
    enlarged by u (a draw from U(1, 2)) times the smallest factor that makes it cover the image in both directions
    (never shrunk below its own size); sizes truncate.  Returns the enlarged (height, width)."""
    (bh, bw), (ih, iw) = bg_hw, im_hw
    cover = max(max(bh, ih) / float(bh), max(bw, iw) / float(bw))
    return int(bh * (u * cover)), int(bw * (u * cover))


def fit_bg(bg, im, rng=rn):
    """A random window of an enlarged copy of `bg`, as large as `im`.  Three draws from `rng`, in this order (so that a
    seeded run picks the windows the reference would): the enlargement u = uniform(1, 2), then the window's top row and
    its left column, each randint over every position that keeps the window inside (both ends included)."""
    ih, iw = im.shape[:2]
    u = rng.uniform(1, 2)
    nh, nw = cover_scale(bg.shape[:2], (ih, iw), u)
    big = np.asarray(Image.fromarray(bg).resize((nw, nh), _ANTIALIAS))
    top = rng.randint(0, big.shape[0] - ih)
    left = rng.randint(0, big.shape[1] - iw)
    return big[top:top + ih, left:left + iw, :]


def add_bg(im, mk, bgim, bgval=0):
    """Composite: pixels whose mask value equals `bgval` come from `bgim`, all others from `im`
    (/root/reference/para_gen.py:50-61).  `im` and `bgim` are (H, W, C) of one shape, `mk` is (H, W)."""
    if im.ndim != 3 or mk.shape != im.shape[:2]:
        raise AssertionError("Sizes mismatch mask and image %s vs. %s" % (mk.shape, im.shape[:-1]))
    if bgim.shape != im.shape:
        raise AssertionError("Sizes mismatch background and image %s vs. %s" % (bgim.shape, im.shape))
    return np.where((mk == bgval)[..., None], bgim, im).astype(im.dtype, copy=False)


MAX_MATCH_DIST = 60          # a match moves less than this many pixels ...


def match_ok(xy1, xy2, msk1, msk2):
    """Which matches become constraints (/root/reference/para_gen.py:216-223)?  xy1, xy2: integer arrays (n, 2) of
    (x, y) in the first / second frame.  A match is kept when both ends lie inside their label masks, it moves by more
    than nothing and by less than MAX_MATCH_DIST pixels (Euclidean; compared as squared integers, which is the same
    predicate), its source lies on a segment (label > 0) and both ends carry the same label.  Returns a bool array.
    (Negative coordinates are rejected here; the reference would index from the far edge, which no matcher output
    can mean.)"""
    xy1 = np.asarray(xy1, np.int64).reshape(-1, 2)
    xy2 = np.asarray(xy2, np.int64).reshape(-1, 2)
    (h1, w1), (h2, w2) = msk1.shape[:2], msk2.shape[:2]
    inside = ((xy1 >= 0).all(1) & (xy2 >= 0).all(1) &
              (xy1[:, 0] < w1) & (xy1[:, 1] < h1) & (xy2[:, 0] < w2) & (xy2[:, 1] < h2))
    d2 = ((xy2 - xy1) ** 2).sum(1)
    keep = inside & (d2 > 0) & (d2 < MAX_MATCH_DIST ** 2)
    lab1 = np.zeros(len(xy1), msk1.dtype)
    lab2 = np.zeros(len(xy1), msk2.dtype)
    lab1[inside] = msk1[xy1[inside, 1], xy1[inside, 0]]
    lab2[inside] = msk2[xy2[inside, 1], xy2[inside, 0]]
    return keep & (lab1 > 0) & (lab1 == lab2)


def valid_cnstr(x1, y1, x2, y2, msk1, msk2):
    """one match (the reference's call shape, para_gen.py:216-223)"""
    return bool(match_ok([(x1, y1)], [(x2, y2)], msk1, msk2)[0])


def filter_matches(match_lines, mk1, mk2):
    """para_gen.py:468-482: the matcher's lines `x1 y1 x2 y2 score index` -> (constraint rows, label of each kept row),
    in the matcher's order"""
    rows = []
    for line in match_lines:
        tok = line.split()
        if len(tok) >= 4:
            rows.append([int(float(t)) for t in tok[:4]])
    if not rows:
        return [], []
    a = np.asarray(rows, np.int64)
    keep = match_ok(a[:, 0:2], a[:, 2:4], mk1, mk2)
    kept = a[keep]
    return [tuple(int(v) for v in r) for r in kept], [int(mk1[r[1], r[0]]) for r in kept]


def write_constraints(path, cstrs):
    """para_gen.py:476-479: count, then tab-separated x1 y1 x2 y2 rows"""
    with open(path, "w") as f:
        f.write("\n".join([str(len(cstrs))] + ["\t".join("%d" % v for v in c) for c in cstrs]))


def resize_crop_geometry(in_size, out_size, margin=10):
    """Frame preparation, size arithmetic (behaviour of /root/reference/para_gen.py:275-287).  in_size, out_size:
    (width, height).  The frame is scaled by the one factor that makes it at least `margin` pixels larger than the
    target in both directions (sizes truncate) and the target window is cut around the centre: left = floor(w / 2) -
    floor(target_w / 2), likewise the top.  Returns ((w, h), (left, upper, right, lower))."""
    (iw, ih), (ow, oh) = in_size, out_size
    r = max((ow + margin) / float(iw), (oh + margin) / float(ih))
    w, h = int(iw * r), int(ih * r)
    left, upper = w // 2 - ow // 2, h // 2 - oh // 2
    return (w, h), (left, upper, left + ow, upper + oh)


def _is_jpeg(path):
    ext = osp.splitext(path)[1].upper()
    return "JPG" in ext or "JPEG" in ext


def scale_rotate(im_path, mk_path, size=None):
    """Open a frame and its label mask and bring them to the working format (/root/reference/para_gen.py:253-291):
    portrait frames are transposed to landscape; with a target `size` (w, h) every frame of another size is scaled
    (image: antialiased, mask: nearest neighbour) and centre-cropped by resize_crop_geometry.  Returns (changed, image,
    mask) as PIL images; `changed` is also true for JPEG inputs, which the caller then re-encodes as PNG."""
    im, mk = Image.open(im_path), Image.open(mk_path)
    if im.size != mk.size:
        raise AssertionError("Image and mask must be of the same size but given %s vs. %s" % (im.size, mk.size))
    changed = _is_jpeg(im_path) or _is_jpeg(mk_path)
    if im.height > im.width:
        im, mk = (x.transpose(Image.TRANSPOSE) for x in (im, mk))
        changed = True
    if size is not None and tuple(im.size) != tuple(size):
        new_size, box = resize_crop_geometry(im.size, tuple(size))
        im = im.resize(new_size, _ANTIALIAS).crop(box)
        mk = mk.resize(new_size, Image.NEAREST).crop(box)
        changed = True
    return changed, im, mk


def make_arap_path(p):
    """para_gen.py:331-339: the list-file line of one solve"""
    return " ".join(osp.abspath(p[k]) for k in ("rgb1_gen", "msk1_gen", "cstr_tmp", "flow_gen", "rgb2_gen", "msk2_gen"))


def replace_ext(dict_path, seg_num, keep_orgs=()):
    """para_gen.py:318-329"""
    out = {}
    for k, v in dict_path.items():
        fn, ext = osp.splitext(v)
        out[k] = v if k in keep_orgs else "%s_seg%d%s" % (fn, seg_num, ext)
    return out


def split_segments(mk1, valid_labels):
    """--multseg (para_gen.py:518-540): one inverted mask per label that has at least one constraint"""
    out = []
    for s in np.unique(valid_labels):
        if s == 0:
            continue
        mask = np.zeros_like(mk1, dtype=np.uint8) + ARAP_BG
        mask[mk1 == s] = 0
        out.append((int(s), mask))
    return out


def merge_segments(flows, rgbs, masks):
    """--multseg: the per-segment results of one frame, stacked (segment, H, W, ...), become one result: at every pixel
    the LAST segment after the first whose warped mask is set wins, and the first segment fills the rest
    (/root/reference/para_gen.py:136-175, which folds the segments in with `old * (mask == 0) + new * (mask != 0)`: the
    same selection, up to the sign of a zero flow and non-finite values under a cleared mask)."""
    masks = np.asarray(masks)
    n = masks.shape[0]
    on = masks != 0
    on[0] = True                                               # the first segment is the base layer
    winner = (n - 1) - np.argmax(on[::-1], axis=0)             # last segment with its mask set
    pick = winner[None, ...]
    flow = np.take_along_axis(np.asarray(flows), pick[..., None], axis=0)[0]
    rgb = np.take_along_axis(np.asarray(rgbs), pick[..., None], axis=0)[0]
    mask = np.take_along_axis(masks, pick, axis=0)[0]
    return flow, rgb, mask


def flatten(arap_seg_paths, remove=True):
    """para_gen.py:136-175 at file level: for every (frame line, [segment lines]) read the segments' flow / warped RGB /
    warped mask files (the last three paths of a list line), merge them (merge_segments), write the frame's three
    files, delete the segments' files.  Returns the frames' list lines."""
    for frame_line, seg_lines in arap_seg_paths:
        if len(seg_lines) == 0:
            raise AssertionError("Something wrong with seg_paths")
        files = [ln.split(" ")[-3:] for ln in seg_lines]
        flows = [flo.flow_read(f) for f, _, _ in files]
        rgbs = [np.asarray(Image.open(r)) for _, r, _ in files]
        rgbs = [a[..., None] if a.ndim == 2 else a for a in rgbs]
        masks = [np.asarray(Image.open(m)) for _, _, m in files]
        flow, rgb, mask = merge_segments(flows, rgbs, masks)
        if remove:
            for trio in files:
                for q in trio:
                    os.remove(q)
        out_flow, out_rgb, out_mask = frame_line.split(" ")[-3:]
        flo.flow_write(out_flow, flow)
        Image.fromarray(rgb.astype(np.uint8).squeeze()).save(out_rgb)
        Image.fromarray(mask.astype(np.uint8)).save(out_mask)     # a 1-bit mask file comes out 0 / 1 valued, as in the reference
    return [e[0] for e in arap_seg_paths]
