"""Python 3 host side of the dataset generator, over libarapopt.so.

Mirrors, function for function, what the reference's Python 2 scripts and C++ drivers do AROUND the hot
path:
  read_list / deform_list   ARAP/deformation/src/main.cpp:162-241  (arap_deform: 6 paths per line)
  warp_files                ARAP/warping/src/main.cpp:302-336      (warp_image)
  fit_bg, add_bg            para_gen.py:36-61
  flatten                   para_gen.py:136-175   (--multseg: merge per-segment outputs by the warped masks)
  valid_cnstr               para_gen.py:216-223
  scale_rotate              para_gen.py:253-291
  make_arap_path            para_gen.py:331-339
No oracle import; the solve and the rasteriser run on the GPU through arap_flow_amd.opt.
"""
import os
import os.path as osp
import random as rn
from math import sqrt

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
def fit_bg(bg, im, rng=rn):
    """para_gen.py:36-48"""
    imh, imw = im.shape[:2]
    bgh, bgw = bg.shape[:2]
    bgim = Image.fromarray(bg)
    hmax, wmax = max(bgh, imh), max(bgw, imw)
    r = rng.uniform(1, 2) * max(float(hmax) / bgh, float(wmax) / bgw)
    bgim = bgim.resize((int(bgw * r), int(bgh * r)), _ANTIALIAS)
    bg = np.array(bgim)
    sy, sx = rng.randint(0, bg.shape[0] - imh), rng.randint(0, bg.shape[1] - imw)
    return bg[sy:(sy + imh), sx:(sx + imw), :]


def add_bg(im, mk, bgim, bgval=0):
    """para_gen.py:50-61"""
    assert mk.shape == im.shape[:-1], "Sizes mismatch mask and image %s vs. %s" % (mk.shape, im.shape[:-1])
    assert bgim.shape == im.shape, "Sizes mismatch background and image %s vs. %s" % (bgim.shape, im.shape)
    out = im.copy()
    idx = mk == bgval
    if len(out.shape) == 3:
        out[idx] = bgim[idx]
    else:
        out = bgim
    return out


def valid_cnstr(x1, y1, x2, y2, msk1, msk2):
    """para_gen.py:216-223: in range, 0 < |d| < 60, on a segment, same label in both masks"""
    if x1 >= msk1.shape[1] or x2 >= msk2.shape[1] or y1 >= msk1.shape[0] or y2 >= msk2.shape[0]:
        return False
    dist = sqrt((x2 - x1) ** 2 + (y2 - y1) ** 2)
    return bool(dist < 60 and dist > 0 and msk1[y1, x1] > 0 and msk1[y1, x1] == msk2[y2, x2])


def filter_matches(match_lines, mk1, mk2):
    """para_gen.py:468-482: keep valid matches; returns (constraint rows, label of each kept row)"""
    cstrs, valids = [], []
    for line in match_lines:
        tok = line.split()
        if len(tok) < 4:
            continue
        x1, y1, x2, y2 = [int(float(t)) for t in tok[:4]]
        if valid_cnstr(x1, y1, x2, y2, mk1, mk2):
            cstrs.append((x1, y1, x2, y2))
            valids.append(int(mk1[y1, x1]))
    return cstrs, valids


def write_constraints(path, cstrs):
    """para_gen.py:476-479: count, then tab-separated x1 y1 x2 y2 rows"""
    with open(path, "w") as f:
        f.write("\n".join([str(len(cstrs))] + ["\t".join("%d" % v for v in c) for c in cstrs]))


def scale_rotate(im_path, mk_path, size=None):
    """para_gen.py:253-291.  Returns (preprocessed, image, mask) as PIL images."""
    im, mk = Image.open(im_path), Image.open(mk_path)
    assert im.size == mk.size, "Image and mask must be of the same size but given %s vs. %s" % (im.size, mk.size)
    ext = "%s %s" % (osp.splitext(im_path)[1], osp.splitext(mk_path)[1])
    preprocessed = "JPG" in ext.upper() or "JPEG" in ext.upper()
    if im.size[1] > im.size[0]:                       # portrait -> landscape
        im, mk = im.transpose(Image.TRANSPOSE), mk.transpose(Image.TRANSPOSE)
        preprocessed = True
    if size is not None and im.size != tuple(size):
        r = max(float(size[0] + 10) / float(im.size[0]), float(size[1] + 10) / float(im.size[1]))
        w, h = (np.array(im.size) * r).astype(int)
        im = im.resize((int(w), int(h)), _ANTIALIAS)
        mk = mk.resize((int(w), int(h)), Image.NEAREST)
        left, upper = int(w / 2) - size[0] // 2, int(h / 2) - size[1] // 2
        box = (left, upper, left + size[0], upper + size[1])
        im, mk = im.crop(box), mk.crop(box)
        preprocessed = True
    return preprocessed, im, mk


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


def flatten(arap_seg_paths, remove=True):
    """para_gen.py:136-175: merge the per-segment flow / warped RGB / warped mask files of each frame, later
    segments overwriting earlier ones where their warped mask is set.  Returns the frames' list lines."""
    for arap_path, seg_paths in arap_seg_paths:
        assert len(seg_paths) > 0, "Something wrong with seg_paths"
        flow_path, rgb2_path, msk2_path = seg_paths[0].split(" ")[-3:]
        flow_im = flo.flow_read(flow_path)
        rgb2_im = np.array(Image.open(rgb2_path))
        msk2_im = np.array(Image.open(msk2_path))
        if rgb2_im.ndim == 2:
            rgb2_im = rgb2_im[..., None]
        if remove:
            for q in (flow_path, rgb2_path, msk2_path):
                os.remove(q)
        for sp in seg_paths[1:]:
            flow_path, rgb2_path, msk2_path = sp.split(" ")[-3:]
            flow_ = flo.flow_read(flow_path)
            rgb2_ = np.array(Image.open(rgb2_path))
            msk2_ = np.array(Image.open(msk2_path))
            msk_ob, msk_bg = msk2_ != 0, msk2_ == 0
            if rgb2_.ndim == 2:
                rgb2_ = rgb2_[..., None]
            flow_im = flow_im * msk_bg[..., None] + flow_ * msk_ob[..., None]
            rgb2_im = rgb2_im * msk_bg[..., None] + rgb2_ * msk_ob[..., None]
            msk2_im = msk2_im * msk_bg + msk2_ * msk_ob
            if remove:
                for q in (flow_path, rgb2_path, msk2_path):
                    os.remove(q)
        out = arap_path.split(" ")
        flo.flow_write(out[-3], flow_im)
        Image.fromarray(rgb2_im.astype(np.uint8).squeeze()).save(out[-2])
        Image.fromarray(msk2_im.astype(np.uint8)).save(out[-1])       # 0/1 valued, as the reference writes it
    return [e[0] for e in arap_seg_paths]
