#!/usr/bin/env python3
"""para_gen -- Python 3 twin of the reference's dataset generator CLI (para_gen.py:341-653), same flags.

  python para_gen.py --input IN --output OUT --gpu 0 1 .. 7 [--fd k] [--size W H] [--multseg] [--resume]
                     [--arap_bin BIN] [--dm_bin BIN | --matches DIR]

Pipeline per frame pair (para_gen.py:384-567): scan IN/orgRGB/**/N.jpg + IN/orgMasks/**/N.png, pair frame n with
n+fd, resize/crop, match, filter matches into constraints, composite a random background, write the inverted
mask (one, or one per segment with --multseg), hand batches of list-file lines to whichever GPU is free (one
child process per GPU, HIP_VISIBLE_DEVICES per child, no collective), flatten segments, composite the
background into the warped frame, write OUT/all_files.list.

Differences from the reference, all at its edges:
  * --arap_bin defaults to this repo's C++ driver arap_flow_amd/bin/arap_deform (arap_deform.py if it is not built);
    any executable with the same argv works.  --narap defaults to 64 (reference: 7): see its comment.
  * DeepMatching (para_gen.py:227-240) is an external binary that is not part of the reference tree.  Either
    pass --dm_bin (called exactly as the reference does) or --matches DIR holding precomputed
    `x1 y1 x2 y2 ...` lines at DIR/<seq>/<frame>.txt.
  * --bg_dir replaces the hard-coded 'data/naturedata' (para_gen.py:16); without it frames keep a black
    background.
"""
import argparse
import logging
import os
import os.path as osp
import random as rn
import re
import subprocess
import sys
import time
from multiprocessing import Process, Queue

import numpy as np
from PIL import Image

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, HERE)
from arap_flow_amd import pipeline          # noqa: E402

orgcolor, orgmask = "orgRGB", "orgMasks"                              # para_gen.py:18-26
color_dir, mask_dir, constraints_dir = "inpRGB", "inpMasks", "tmpCnstr"
flow_dir, wrgb_dir, wMask_dir = "Flow", "wRGB", "wMasks"


def run_matching(flags, p, seq, stem):
    """para_gen.py:227-240, or precomputed matches"""
    for k in ("rgb1_org", "rgb2_org", "msk1_org", "msk2_org"):
        assert osp.exists(p[k]), "File not found: \n%s" % p[k]
    if flags.matches is not None:
        src = osp.join(flags.matches, seq, stem + ".txt")
        assert osp.exists(src), "File not found: \n%s" % src
        open(p["cstr_tmp"], "w").write(open(src).read())
        return
    cmd = "./%s %s %s -nt 0 -out %s -ngh_rad 100 " % (flags.dm_bin, p["rgb1_org"], p["rgb2_org"], p["cstr_tmp"])
    status = subprocess.call(cmd, shell=True)
    assert status == 0, "Deep matching exited with code %d. The command is \n%s" % (status, cmd)


def has_mask(m1, m2):
    """para_gen.py:243-251"""
    try:
        a, b = np.array(Image.open(m1)), np.array(Image.open(m2))
    except Exception:
        return False
    return a.sum() > 10 and b.sum() > 10


def preprocess(p, size):
    """para_gen.py:294-310"""
    out = []
    for n in ("1", "2"):
        pre, im, mk = pipeline.scale_rotate(p["rgb%s_org" % n], p["msk%s_org" % n], size)
        if pre:
            im.save(p["rgb%s_gen" % n])
            mk.save(p["msk%s_gen" % n])
            p["rgb%s_org" % n], p["msk%s_org" % n] = p["rgb%s_gen" % n], p["msk%s_gen" % n]
        out += [np.array(im.convert("RGB")), np.array(mk)]
    return out


def cleanup(p):
    """para_gen.py:311-316"""
    for k in p:
        if "_org" not in k and osp.exists(p[k]):
            logging.warning("Removing\n\t%s", p[k])
            os.remove(p[k])


def do_arap(flags, paths, bgs, gpu, gpu_queue, arap_seg_paths):
    """para_gen.py:178-214: one child per GPU; list file -> arap binary; flatten; background"""
    os.makedirs("tmp", exist_ok=True)
    fn = osp.abspath("tmp/gpu-%d_%s.txt" % (gpu, str(time.time()).replace(".", "_")))
    print("GPU ", gpu, " ", len(paths), " files")
    try:
        open(fn, "w").write("\n".join(paths))
        cmd = flags.arap_bin.split() + [fn]
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(gpu))           # reference: CUDA_VISIBLE_DEVICES (:190)
        status = subprocess.call(cmd, env=env)
        assert status == 0, "ARAP exited with code %d. The command was \n%s" % (status, " ".join(cmd))
    finally:
        os.remove(fn)
    if len(arap_seg_paths) > 0:
        paths = pipeline.flatten(arap_seg_paths)
    for path, bg in zip(paths, bgs):
        if bg is None:
            continue
        pt, mk = path.split(" ")[-2:]
        im = np.array(Image.open(pt).convert("RGB"))
        m = np.array(Image.open(mk))
        Image.fromarray(pipeline.add_bg(im, m, bg)).save(pt)
    gpu_queue.put(gpu)


def scan(flags, input_root, output_root):
    """para_gen.py:384-432"""
    rgb_org, msk_org = osp.join(input_root, orgcolor), osp.join(input_root, orgmask)
    roots = {k: osp.join(output_root, v) for k, v in dict(cst=constraints_dir, flo=flow_dir, rgb=color_dir,
                                                           msk=mask_dir, wco=wrgb_dir, wmk=wMask_dir).items()}
    reg = re.compile(r"(\d+)\.(jp.?g|png)$", flags=re.IGNORECASE)
    all_paths = []
    for root, dirs, _ in os.walk(rgb_org):
        for d in sorted(dirs):
            files = sorted(f for f in os.listdir(osp.join(root, d)) if reg.search(f) is not None)
            for f1 in files:
                seq = osp.join(root.replace(rgb_org, "").strip(osp.sep), d)
                f, ext = osp.splitext(f1)
                if not osp.exists(osp.join(msk_org, seq, f + ".png")):
                    continue
                num = reg.search(f1)
                n = "{:0" + str(len(num.group(1))) + "d}"
                f2 = f.replace(num.group(1), n.format(int(num.group(1)) + flags.fd))
                if not osp.exists(osp.join(rgb_org, seq, f2 + ext)) or not osp.exists(osp.join(msk_org, seq, f2 + ".png")):
                    continue
                e = dict(rgb1_gen=osp.join(roots["rgb"], seq, f + ".png"), msk1_gen=osp.join(roots["msk"], seq, f + ".png"),
                         rgb2_gen=osp.join(roots["wco"], seq, f + ".png"), msk2_gen=osp.join(roots["wmk"], seq, f + ".png"),
                         cstr_tmp=osp.join(roots["cst"], seq, f + ".txt"), flow_gen=osp.join(roots["flo"], seq, f + ".flo"),
                         rgb1_org=osp.join(rgb_org, seq, f1), msk1_org=osp.join(msk_org, seq, f + ".png"),
                         rgb2_org=osp.join(rgb_org, seq, f2 + ext), msk2_org=osp.join(msk_org, seq, f2 + ".png"))
                e = {k: osp.abspath(v) for k, v in e.items()}
                e["_seq"], e["_stem"] = seq, f
                if not flags.resume or not osp.exists(e["flow_gen"]):            # --resume (:431)
                    all_paths.append(e)
    return all_paths


def main(flags):
    input_root, output_root = flags.input.rstrip(osp.sep), flags.output.rstrip(osp.sep)
    bg_paths = []
    if flags.bg_dir:
        for root, _, files in os.walk(flags.bg_dir):
            bg_paths += [osp.join(root, f) for f in files if f.upper().endswith((".PNG", ".JPG", ".JPEG"))]
    tmp_paths = []
    all_paths = scan(flags, input_root, output_root)
    print("Scanning data to be processed\t\t%d files [Done]" % len(all_paths))
    lmdb_paths, arap_paths, arap_seg_paths, bgs, procs = [], [], [], [], {}
    gpu_queue = Queue(len(flags.gpu))
    for g in flags.gpu:
        gpu_queue.put(g)

    def dispatch(block):
        nonlocal arap_paths, arap_seg_paths, bgs
        if not arap_paths or (gpu_queue.empty() and not block):
            return
        gpu = gpu_queue.get()
        proc = Process(target=do_arap, args=(flags, arap_paths, bgs, gpu, gpu_queue, arap_seg_paths))
        proc.start()
        procs.setdefault(gpu, []).append(proc)
        arap_paths, arap_seg_paths, bgs = [], [], []

    for i, p in enumerate(all_paths):
        print("%.3f%%" % (float(i) * 100 / len(all_paths)))
        seq, stem = p.pop("_seq"), p.pop("_stem")
        arap_path = pipeline.make_arap_path(p)
        ap = arap_path.split(" ")
        lmdb_paths.append(" ".join([ap[0], ap[4], ap[3]]))
        for k in p:
            os.makedirs(osp.dirname(p[k]), exist_ok=True)
        im1, mk1, im2, mk2 = preprocess(p, flags.size)
        if not has_mask(p["msk1_org"], p["msk2_org"]):
            cleanup(p)
            continue
        run_matching(flags, p, seq, stem)
        cstr_lines = open(p["cstr_tmp"]).read().splitlines()
        cstrs, valids = pipeline.filter_matches(cstr_lines, mk1, mk2)
        pipeline.write_constraints(p["cstr_tmp"], cstrs)
        if len(cstrs) == 0:
            cleanup(p)
            continue
        bgim = None
        while bg_paths:
            if len(tmp_paths) == 0:
                tmp_paths = sorted(bg_paths[:])
            bgpath = rn.choice(tmp_paths)
            tmp_paths.remove(bgpath)
            try:
                bgim = np.array(Image.open(bgpath))
                if bgim.ndim == 3 and bgim.shape[2] == 3:
                    break
            except Exception:
                pass
            bg_paths.remove(bgpath)
            bgim = None
        if bgim is not None:
            bgim = pipeline.fit_bg(bgim, im1)
            out1 = pipeline.add_bg(im1, mk1, bgim)
        else:
            out1 = im1
        bgs.append(bgim)
        Image.fromarray(out1).save(p["rgb1_gen"])
        seg_paths = None
        if not flags.multseg:
            mask = np.zeros_like(mk1, dtype=np.uint8)
            mask[mk1 == 0] = pipeline.ARAP_BG                                      # :514-517
            Image.fromarray(mask).save(p["msk1_gen"])
        else:
            seg_paths = []
            for s, mask in pipeline.split_segments(mk1, valids):                   # :518-540
                p_ = pipeline.replace_ext(p, s, keep_orgs=["rgb1_gen", "cstr_tmp"])
                Image.fromarray(mask).save(p_["msk1_gen"])
                seg_paths.append(pipeline.make_arap_path(p_))
            arap_seg_paths.append((arap_path, seg_paths))
        arap_paths += [arap_path] if seg_paths is None else seg_paths
        dispatch(block=False)                                                      # :560-567
    while arap_paths:
        dispatch(block=True)
    for lst in procs.values():
        for proc in lst:
            proc.join()
            assert proc.exitcode == 0, "ARAP worker failed"
    out_paths = [ln for ln in lmdb_paths if all(osp.exists(q) for q in ln.split(" "))]   # :588-603
    open(osp.join(output_root, "all_files.list"), "w").write("\n".join(out_paths))
    return out_paths


def parse(argv=None):
    parser = argparse.ArgumentParser(description="Arguments for ARAP flow generation")
    parser.add_argument("--input", type=str, required=True, help="Path to input root")
    parser.add_argument("--output", type=str, required=True, help="Path to output root")
    parser.add_argument("--rm-cnstr")
    parser.add_argument("--rm-wmask")
    parser.add_argument("--rm-tmp-cmd")
    parser.add_argument("--img-pattern")
    parser.add_argument("--gpu", nargs="*", type=int, default=[0], help="GPU id to be used, default=0")
    parser.add_argument("--multseg", action="store_true", default=False,
                        help="if each object segment is treated separately")
    parser.add_argument("--resume", action="store_true", default=False,
                        help="To skip the images that have *.flo finished.")
    # (reference default: 7.  A child here costs ~0.4 s to start and solves 8 854x480 frames per 0.4 s launch, so
    #  larger hand-outs keep the GPUs busy; any value works.)
    parser.add_argument("--narap", type=int, default=64, help="Number of buffered files to be run by ARAP on gpu")
    parser.add_argument("--size", nargs=2, default=None,
                        help="2-tuple of [width] [space] [height] to which all images are resized.")
    parser.add_argument("--fd", type=int, default=1, help="distance between the 2 frames, default=1")
    cpp_bin = osp.join(HERE, "arap_flow_amd", "bin", "arap_deform")
    parser.add_argument("--arap_bin", default=cpp_bin if osp.exists(cpp_bin) else "%s %s" % (sys.executable, osp.join(HERE, "arap_deform.py")),
                        help="ARAP executable (argv contract of arap_deform), default: this repo's C++ driver "
                             "arap_flow_amd/bin/arap_deform when it is built, else arap_deform.py")
    parser.add_argument("--dm_bin", default=None, help="Path to the deep matching binary")
    parser.add_argument("--matches", default=None, help="directory of precomputed matches (instead of --dm_bin)")
    parser.add_argument("--bg_dir", default=None, help="directory of background images")
    flags = parser.parse_args(argv)
    if flags.size is not None:
        flags.size = tuple(int(s) for s in flags.size)
    assert 0 < flags.fd < 20, "Invalid fd number!"
    assert flags.dm_bin is not None or flags.matches is not None, "give --dm_bin or --matches"
    if flags.dm_bin is not None:
        assert osp.exists(flags.dm_bin), "File not found " + flags.dm_bin
    return flags


if __name__ == "__main__":
    logging.basicConfig(filename="example.log", level=logging.DEBUG)
    main(parse())
