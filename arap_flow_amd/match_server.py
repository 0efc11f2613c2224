"""Matcher server: one process per GPU that holds the HIP context of libarapmatch.so and matches frame pairs on request.

    python -m arap_flow_amd.match_server SOCKET_PATH          (HIP_VISIBLE_DEVICES selects the GPU)

`para_gen.py --dm_bin builtin` starts one per --gpu id; its front-end worker processes (which must not each open a
HIP context of their own) send (frame 1 path, frame 2 path, output path, ngh_rad) over a Unix socket and get the
number of matches back; the server writes the output file in the format of the binary it stands in for
(`x1 y1 x2 y2 score index` per line, /root/reference/para_gen.py:227-240,468-479)."""
import os
import sys
from multiprocessing.connection import Client, Listener

AUTH = b"arap-match"


def request(sock_path, img1, img2, out_path, ngh_rad=100):
    """client side (any process): blocks until the pair is matched and out_path is written; returns the match count"""
    with Client(sock_path, family="AF_UNIX", authkey=AUTH) as c:
        c.send((img1, img2, out_path, int(ngh_rad)))
        ok, val = c.recv()
    if not ok:
        raise RuntimeError("matcher: %s" % val)
    return val


def serve(sock_path):
    import numpy as np
    from PIL import Image
    from . import match
    match.load()
    matchers = {}
    if os.path.exists(sock_path):
        os.remove(sock_path)
    with Listener(sock_path, family="AF_UNIX", authkey=AUTH) as ls:
        print("Ready", flush=True)
        while True:
            with ls.accept() as c:
                msg = c.recv()
                if msg == "quit":
                    c.send((True, 0))
                    return
                try:
                    img1, img2, out_path, rad = msg
                    a = np.array(Image.open(img1).convert("RGB"))
                    b = np.array(Image.open(img2).convert("RGB"))
                    if a.shape != b.shape:
                        raise ValueError("frames differ in size: %s vs %s" % (a.shape, b.shape))
                    key = (a.shape[1], a.shape[0], rad)
                    if key not in matchers:
                        matchers[key] = match.Matcher(*key)
                    m = matchers[key].run(a, b)
                    with open(out_path, "w") as f:
                        f.write("\n".join(match.format_lines(m)))
                    c.send((True, len(m)))
                except Exception as e:                           # the client raises; the server lives on
                    c.send((False, "%s: %s" % (type(e).__name__, e)))


def stop(sock_path):
    try:
        with Client(sock_path, family="AF_UNIX", authkey=AUTH) as c:
            c.send("quit")
            c.recv()
    except OSError:
        pass


if __name__ == "__main__":
    serve(sys.argv[1])
