#!/usr/bin/env python3
"""para_gen -- Python 3 twin of the reference's dataset generator CLI (para_gen.py:341-653), same flags.

  python para_gen.py --input IN --output OUT --gpu 0 1 .. 7 [--fd k] [--size W H] [--multseg] [--resume]
                     [--arap_bin BIN] [--dm_bin BIN | --matches DIR] [--narap N] [--jobs J]

Pipeline per frame pair (para_gen.py:384-567): scan IN/orgRGB/**/N.jpg + IN/orgMasks/**/N.png, pair frame n with
n+fd, resize/crop, match, filter matches into constraints, composite a random background, write the inverted
mask (one, or one per segment with --multseg), hand list-file lines to whichever GPU has room (one worker process
per GPU, HIP_VISIBLE_DEVICES per worker, no collective), flatten segments, composite the background into the
warped frame, write OUT/all_files.list.

What differs from the reference, and why (a GPU solves ~20 of these frames per second; the reference's loop prepares
pairs serially and forks one ARAP child per hand-out, which would leave the GPU idle > 90 % of the time):
  * The front end (resize, match filter, masks, PNG writes) and the back end (flatten, background) run in --jobs worker
    processes; the parent only schedules.  Output files are the same.
  * One persistent ARAP worker per GPU id (`arap_deform --serve`, started before the first pair is prepared) is fed
    list-file lines over a pipe and reports every finished solve; it batches what has accumulated whenever its GPU
    would otherwise idle.  --narap is the most lines a GPU holds at a time (the reference's dead flag, para_gen.py:628,
    put to work): lines go to whichever GPU has room, as the reference's GPU queue does (para_gen.py:441-445,560-567).
    A foreign --arap_bin (argv contract of arap_deform only) is run once per hand-out of up to --narap lines instead.
  * A worker that exits non-zero fails the run at once (the reference hangs: its GPU id never returns to the queue).
  * DeepMatching (para_gen.py:227-240) is an external binary that is not part of the reference tree.  Pass --dm_bin
    PATH (called exactly as the reference does), --dm_bin builtin (this repo's GPU implementation of the published
    algorithm, libarapmatch.so: every pair is matched first, then the ARAP workers start -- the two must not share a
    GPU at the same time) or --matches DIR holding precomputed `x1 y1 x2 y2 ...` lines at DIR/<seq>/<frame>.txt.
  * --bg_dir replaces the hard-coded 'data/naturedata' (para_gen.py:16); without it frames keep a black
    background.
"""
import argparse
import json
import logging
import os
import os.path as osp
import queue
import random as rn
import re
import subprocess
import sys
import threading
import time
from multiprocessing import Pool

import numpy as np
from PIL import Image

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, HERE)
from arap_flow_amd import pipeline          # noqa: E402

orgcolor, orgmask = "orgRGB", "orgMasks"                              # para_gen.py:18-26
color_dir, mask_dir, constraints_dir = "inpRGB", "inpMasks", "tmpCnstr"
flow_dir, wrgb_dir, wMask_dir = "Flow", "wRGB", "wMasks"
CPP_BIN = osp.join(HERE, "arap_flow_amd", "bin", "arap_deform")


def _pair_id(seq, stem):
    """a number per frame pair that is the same in every process and every run (str hashes are salted per process)"""
    import zlib
    return zlib.crc32(("%s/%s" % (seq, stem)).encode())


def run_matching(flags, p, seq, stem):
    """para_gen.py:227-240, or precomputed matches"""
    for k in ("rgb1_org", "rgb2_org", "msk1_org", "msk2_org"):
        assert osp.exists(p[k]), "File not found: \n%s" % p[k]
    if flags.matches is not None:
        src = osp.join(flags.matches, seq, stem + ".txt")
        assert osp.exists(src), "File not found: \n%s" % src
        open(p["cstr_tmp"], "w").write(open(src).read())
        return
    if flags.dm_bin == "builtin":
        # this repo's matcher (libarapmatch.so, include/arap_match.h) through the matcher server of one of the GPUs: same
        # inputs, same -ngh_rad, same output file format as the binary below
        from arap_flow_amd import match_server
        socks = flags.dm_sockets
        match_server.request(socks[_pair_id(seq, stem) % len(socks)], p["rgb1_org"], p["rgb2_org"], p["cstr_tmp"], 100)
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


# ----------------------------------------------------------------------------------------------------------------
# front end / back end of one frame pair: run in the --jobs pool
# ----------------------------------------------------------------------------------------------------------------
def prepare_pair(args):
    """para_gen.py:447-556 for one pair: everything up to its list-file line(s).  Returns None when the pair is
    dropped (no mask, no valid constraint), else dict(arap_path, seg_paths or None, bg)."""
    flags, p, bgpath = args
    p = dict(p)
    seq, stem = p.pop("_seq"), p.pop("_stem")
    arap_path = pipeline.make_arap_path(p)
    for k in p:
        os.makedirs(osp.dirname(p[k]), exist_ok=True)
    im1, mk1, im2, mk2 = preprocess(p, flags.size)
    if not has_mask(p["msk1_org"], p["msk2_org"]):
        cleanup(p)
        return None
    run_matching(flags, p, seq, stem)
    cstr_lines = open(p["cstr_tmp"]).read().splitlines()
    cstrs, valids = pipeline.filter_matches(cstr_lines, mk1, mk2)
    pipeline.write_constraints(p["cstr_tmp"], cstrs)
    if len(cstrs) == 0:
        cleanup(p)
        return None
    bgim = None
    if bgpath is not None:
        try:
            bgim = np.array(Image.open(bgpath))
            if not (bgim.ndim == 3 and bgim.shape[2] == 3):
                bgim = None
        except Exception:
            bgim = None
    if bgim is not None:
        bgim = pipeline.fit_bg(bgim, im1, rng=rn.Random(_pair_id(seq, stem)))
        out1 = pipeline.add_bg(im1, mk1, bgim)
    else:
        out1 = im1
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
    return dict(arap_path=arap_path, seg_paths=seg_paths, bg=bgim)


def finish_frame(args):
    """para_gen.py:202-212 for one frame whose solve(s) are done: flatten the segments, composite the background"""
    arap_path, seg_paths, bg = args
    if seg_paths is not None:
        pipeline.flatten([(arap_path, seg_paths)])
    if bg is not None:
        pt, mk = arap_path.split(" ")[-2:]
        im = np.array(Image.open(pt).convert("RGB"))
        m = np.array(Image.open(mk))
        Image.fromarray(pipeline.add_bg(im, m, bg)).save(pt)
    return arap_path


# ----------------------------------------------------------------------------------------------------------------
# GPU side: one worker per GPU id, lines to whichever has room
# ----------------------------------------------------------------------------------------------------------------
class GpuWorkers:
    """The reference's GPU queue (para_gen.py:441-445,560-567) at line granularity.  `serve`: one persistent
    `arap_bin --serve` child per GPU; `batch`: one `arap_bin listfile` child per hand-out of up to `narap` lines."""

    def __init__(self, arap_bin, gpus, narap, serve, on_done):
        self.cmd, self.narap, self.serve, self.on_done = arap_bin.split(), max(1, int(narap)), serve, on_done
        self.lines = queue.Queue()
        self.error = None
        self.batches = []                      # solves per GPU launch (serve) / per child (batch)
        self.threads, self.procs = [], []
        self.t_ready = None
        for g in gpus:
            t = threading.Thread(target=self._serve_loop if serve else self._batch_loop, args=(g,), daemon=True)
            t.start()
            self.threads.append(t)

    def put(self, line):
        self.lines.put(line)

    def close(self):
        for _ in self.threads:
            self.lines.put(None)

    def join(self):
        for t in self.threads:
            while t.is_alive():
                t.join(0.2)
                self.check()
        self.check()

    def check(self):
        if self.error is not None:
            for p in self.procs:
                if p.poll() is None:
                    p.kill()
            raise AssertionError(self.error)

    def _fail(self, msg):
        if self.error is None:
            self.error = msg

    def _env(self, gpu):
        return dict(os.environ, HIP_VISIBLE_DEVICES=str(gpu))                  # reference: CUDA_VISIBLE_DEVICES (:190)

    # -- persistent worker -----------------------------------------------------------------------------------
    def _serve_loop(self, gpu):
        try:
            proc = subprocess.Popen(self.cmd + ["--serve"], env=self._env(gpu), stdin=subprocess.PIPE,
                                    stdout=subprocess.PIPE, text=True, bufsize=1)
        except OSError as e:
            self._fail("cannot start %s: %s" % (" ".join(self.cmd), e))
            return
        self.procs.append(proc)
        room = threading.Semaphore(self.narap)
        outstanding = [0]
        lock = threading.Lock()

        def reader():
            for ln in proc.stdout:
                ln = ln.rstrip("\n")
                if ln.startswith("Done "):
                    with lock:
                        outstanding[0] -= 1
                    room.release()
                    self.on_done(ln[5:])
                elif ln.startswith("Batch "):
                    self.batches.append(int(ln[6:]))
                elif ln == "Ready":
                    if self.t_ready is None:
                        self.t_ready = time.time()
                elif ln and ln != "Saved":
                    print(ln)
            rc = proc.wait()
            with lock:
                left = outstanding[0]
            if rc != 0 or left != 0:
                self._fail("ARAP worker on GPU %d exited with code %d, %d solves unfinished. The command was \n%s"
                           % (gpu, rc, left, " ".join(self.cmd + ["--serve"])))
            room.release()                                                      # never leave the feeder blocked

        rt = threading.Thread(target=reader, daemon=True)
        rt.start()
        while self.error is None:
            room.acquire()                                                      # a free place on this GPU first ...
            if proc.poll() is not None:
                break
            line = self.lines.get()                                             # ... then the next line, whoever it is
            if line is None:
                break
            with lock:
                outstanding[0] += 1
            try:
                proc.stdin.write(line + "\n")
                proc.stdin.flush()
            except (BrokenPipeError, OSError):
                self._fail("ARAP worker on GPU %d closed its input" % gpu)
                break
        try:
            proc.stdin.close()
        except OSError:
            pass
        rt.join()

    # -- one child per hand-out (any executable with arap_deform's argv contract) -------------------------------
    def _batch_loop(self, gpu):
        os.makedirs("tmp", exist_ok=True)
        while self.error is None:
            line = self.lines.get()
            if line is None:
                return
            batch, last = [line], False
            while len(batch) < self.narap:
                try:
                    nxt = self.lines.get(timeout=0.05)
                except queue.Empty:
                    break
                if nxt is None:
                    last = True
                    break
                batch.append(nxt)
            fn = osp.abspath("tmp/gpu-%d_%s.txt" % (gpu, str(time.time()).replace(".", "_")))
            print("GPU ", gpu, " ", len(batch), " files")
            try:
                open(fn, "w").write("\n".join(batch))
                status = subprocess.call(self.cmd + [fn], env=self._env(gpu))
            except OSError as e:
                status = "not started (%s)" % e
            finally:
                if osp.exists(fn):
                    os.remove(fn)
            if status != 0:
                self._fail("ARAP exited with code %s. The command was \n%s" % (status, " ".join(self.cmd + [fn])))
                return
            self.batches.append(len(batch))
            for ln in batch:
                self.on_done(ln.split(" ")[3])
            if last:
                return


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


def start_matchers(flags, output_root):
    """one matcher server per --gpu id (arap_flow_amd/match_server.py); their socket paths go to the pool jobs in flags"""
    import tempfile
    d = tempfile.mkdtemp(prefix="arapmatch_")
    procs, socks = [], []
    for g in flags.gpu:
        sock = osp.join(d, "gpu%d.sock" % g)
        env = dict(os.environ, HIP_VISIBLE_DEVICES=str(g))
        pr = subprocess.Popen([sys.executable, "-m", "arap_flow_amd.match_server", sock], cwd=HERE, env=env,
                              stdout=subprocess.PIPE, text=True)
        line = pr.stdout.readline()
        assert line.strip() == "Ready", "matcher server on GPU %d did not start" % g
        procs.append(pr)
        socks.append(sock)
    flags.dm_sockets = socks
    return procs


def stop_matchers(flags, procs):
    from arap_flow_amd import match_server
    for sock in getattr(flags, "dm_sockets", []):
        match_server.stop(sock)
    for pr in procs:
        try:
            pr.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pr.kill()


def main(flags):
    t_start = time.time()
    input_root, output_root = flags.input.rstrip(osp.sep), flags.output.rstrip(osp.sep)
    bg_paths = []
    if flags.bg_dir:
        for root, _, files in os.walk(flags.bg_dir):
            bg_paths += [osp.join(root, f) for f in files if f.upper().endswith((".PNG", ".JPG", ".JPEG"))]
    all_paths = scan(flags, input_root, output_root)
    print("Scanning data to be processed\t\t%d files [Done]" % len(all_paths))
    os.makedirs(output_root, exist_ok=True)
    lmdb_paths = []
    for p in all_paths:
        q = {k: v for k, v in p.items() if not k.startswith("_")}
        ap = pipeline.make_arap_path(q).split(" ")
        lmdb_paths.append(" ".join([ap[0], ap[4], ap[3]]))

    # backgrounds: drawn without replacement until the list is used up, then refilled (para_gen.py:484-499)
    tmp_paths, picks = [], []
    for _ in all_paths:
        if not bg_paths:
            picks.append(None)
            continue
        if not tmp_paths:
            tmp_paths = sorted(bg_paths[:])
        bgpath = rn.choice(tmp_paths)
        tmp_paths.remove(bgpath)
        picks.append(bgpath)

    serve = flags.worker == "serve" or (flags.worker == "auto" and osp.abspath(flags.arap_bin.split()[0]) == CPP_BIN)
    pool = Pool(processes=max(1, flags.jobs))          # (forked before any thread exists)
    frames = {}                                        # flow path of a solve -> its frame record
    posts, lock = [], threading.Lock()
    counts = dict(solves_done=0, frames_done=0)

    def on_done(flow_path):                            # a worker thread: one solve finished
        with lock:
            rec = frames.pop(flow_path)
            counts["solves_done"] += 1
            rec["left"] -= 1
            if rec["left"] > 0:
                return
            counts["frames_done"] += 1
            posts.append(pool.apply_async(finish_frame, ((rec["arap_path"], rec["seg_paths"], rec["bg"]),)))

    # --dm_bin builtin: the matcher and the solver must not share a GPU at the same time (the solver's resident kernel
    # needs the whole chip: arap_resident.h), so the run has two phases -- every pair is prepared and matched first
    # (one matcher server per GPU id), the servers exit, then the ARAP workers start and are fed.
    prepared = None
    if flags.dm_bin == "builtin":
        servers = start_matchers(flags, output_root)
        try:
            jobs = ((flags, p, bg) for p, bg in zip(all_paths, picks))
            prepared = list(pool.imap(prepare_pair, jobs, chunksize=1))
        finally:
            stop_matchers(flags, servers)
        print("Matching		%d pairs [Done]" % len(prepared))
    workers = GpuWorkers(flags.arap_bin, flags.gpu, flags.narap, serve, on_done)
    n_solves = n_frames = 0
    try:
        jobs = ((flags, p, bg) for p, bg in zip(all_paths, picks))
        for i, res in enumerate(prepared if prepared is not None else pool.imap(prepare_pair, jobs, chunksize=1)):
            print("%.3f%%" % (float(i) * 100 / len(all_paths)))
            workers.check()
            if res is None:
                continue
            lines = [res["arap_path"]] if res["seg_paths"] is None else res["seg_paths"]
            if not lines:
                continue
            rec = dict(arap_path=res["arap_path"], seg_paths=res["seg_paths"], bg=res["bg"], left=len(lines))
            with lock:
                for ln in lines:
                    frames[ln.split(" ")[3]] = rec
            for ln in lines:
                workers.put(ln)
            n_solves += len(lines)
            n_frames += 1
        workers.close()
        workers.join()
        for r in posts:
            r.get()
    finally:
        pool.terminate()
        for p in workers.procs:
            if p.poll() is None:
                p.kill()
    out_paths = [ln for ln in lmdb_paths if all(osp.exists(q) for q in ln.split(" "))]   # :588-603
    open(osp.join(output_root, "all_files.list"), "w").write("\n".join(out_paths))
    dt = time.time() - t_start
    stats = dict(pairs=len(all_paths), frames=n_frames, solves=n_solves, seconds=dt,
                 seconds_since_workers_ready=(time.time() - workers.t_ready) if workers.t_ready else None,
                 gpus=list(flags.gpu), worker="serve" if serve else "batch", jobs=flags.jobs, narap=flags.narap,
                 batches=workers.batches,
                 mean_batch=(float(np.mean(workers.batches)) if workers.batches else 0.0))
    open(osp.join(output_root, "arap_stats.json"), "w").write(json.dumps(stats))
    print("Finished: %d frames (%d solves) in %.2f s, mean batch %.1f" % (n_frames, n_solves, dt, stats["mean_batch"]))
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
    # reference default: 7 (and never read there).  Here: the most list lines one GPU holds at a time -- enough for the
    # batch being solved, the one being uploaded and the one being assembled (8 854x480 frames or ~21 segments each).
    parser.add_argument("--narap", type=int, default=64, help="Number of buffered files to be run by ARAP on gpu")
    parser.add_argument("--size", nargs=2, default=None,
                        help="2-tuple of [width] [space] [height] to which all images are resized.")
    parser.add_argument("--fd", type=int, default=1, help="distance between the 2 frames, default=1")
    parser.add_argument("--arap_bin", default=CPP_BIN if osp.exists(CPP_BIN) else "%s %s" % (sys.executable, osp.join(HERE, "arap_deform.py")),
                        help="ARAP executable (argv contract of arap_deform), default: this repo's C++ driver "
                             "arap_flow_amd/bin/arap_deform when it is built, else arap_deform.py")
    parser.add_argument("--worker", choices=["auto", "serve", "batch"], default="auto",
                        help="serve: one persistent `arap_bin --serve` per GPU (this repo's C++ driver); batch: one "
                             "`arap_bin listfile` child per hand-out; auto: serve for this repo's driver")
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    try:                                                   # a container's CPU quota (cgroup v2), if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            ncpu = max(1, min(ncpu, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    parser.add_argument("--jobs", type=int, default=max(1, min(96, ncpu - 2)),
                        help="worker processes for the per-pair front end and back end")
    parser.add_argument("--dm_bin", default=None,
                        help="Path to the deep matching binary, or 'builtin': this repo's GPU matcher (libarapmatch.so)")
    parser.add_argument("--matches", default=None, help="directory of precomputed matches (instead of --dm_bin)")
    parser.add_argument("--bg_dir", default=None, help="directory of background images")
    flags = parser.parse_args(argv)
    if flags.size is not None:
        flags.size = tuple(int(s) for s in flags.size)
    assert 0 < flags.fd < 20, "Invalid fd number!"
    assert flags.dm_bin is not None or flags.matches is not None, "give --dm_bin or --matches"
    if flags.dm_bin is not None and flags.dm_bin != "builtin":
        assert osp.exists(flags.dm_bin), "File not found " + flags.dm_bin
    return flags


if __name__ == "__main__":
    logging.basicConfig(filename="example.log", level=logging.DEBUG)
    main(parse())
