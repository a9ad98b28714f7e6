"""Host -> HBM upload rate of the library's uploader (csrc/fast_upload.hpp) from a page-cached file, alone and beside
compute -- the path every zkey section and every witness takes. One sub-process per setting (the
uploader reads ZKPOA_UPLOAD_THREADS / _STREAMS / _AFFINITY once).

  python tools/upload_bench.py [--gb 8] [--dir /dev/shm]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path, nbytes):
    import numpy as np
    import torch
    from __graft_entry__ import load_package
    z = load_package()
    ctx = z.Context(0)
    L = z.lib()
    L.zkpoa_test_upload.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p,
                                    ctypes.POINTER(ctypes.c_float)]
    dst = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    ms = ctypes.c_float(0)

    def up():
        rc = L.zkpoa_test_upload(ctx._h, path.encode(), 0, nbytes, dst.data_ptr(), ctypes.byref(ms))
        assert rc == 0
        return nbytes / 1e6 / ms.value     # GB/s

    time.sleep(0.5)                        # the uploader's own streams come up in the background
    up()
    idle = [up() for _ in range(4)]
    # the same beside compute: four lanes running 2^22-point MSMs
    n = 1 << 22
    bases = torch.empty(n * 64, dtype=torch.uint8, device="cuda")
    ctx.gen_bases_g1_device(3, 5, 0, n, bases.data_ptr())
    sc = torch.from_numpy(np.random.default_rng(1).integers(0, 1 << 62, size=(n, 4), dtype=np.uint64).view(np.uint8).reshape(-1)).cuda()
    stop = threading.Event()

    def load(lane):
        while not stop.is_set():
            ctx.msm_g1_device_lane(lane, bases.data_ptr(), sc.data_ptr(), n)
    th = [threading.Thread(target=load, args=(l,)) for l in (1, 2, 3, 4)]
    for t in th:
        t.start()
    time.sleep(0.5)
    busy = [up() for _ in range(4)]
    stop.set()
    for t in th:
        t.join()
    ok = bool((dst[:1 << 20].cpu().numpy() == np.fromfile(path, dtype=np.uint8, count=1 << 20)).all())
    print(json.dumps({"idle_GBs": [round(x, 1) for x in idle], "beside_4_msm_lanes_GBs": [round(x, 1) for x in busy], "bytes_ok": ok}))
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=8.0)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--child", default=None)
    args = ap.parse_args()
    nbytes = int(args.gb * 1e9) & ~0xfff
    if args.child:
        return child(args.child, nbytes)
    import numpy as np
    path = os.path.join(args.dir, "zkpoa_upload_bench.bin")
    blk = np.random.default_rng(0).integers(0, 255, size=1 << 26, dtype=np.uint8)
    with open(path, "wb") as f:
        left = nbytes
        while left > 0:
            f.write(memoryview(blk[:min(left, blk.size)]))
            left -= blk.size
    try:
        variants = [(8, 0, {}), (8, 1, {}), (8, 3, {}), (8, 1, {"ZKPOA_UPLOAD_STREAM_PRIO": "high"}),
                    (8, 3, {"ZKPOA_UPLOAD_STREAM_PRIO": "high"}),
                    (8, 1, {"GPU_MAX_HW_QUEUES": "8"}), (8, 1, {"GPU_MAX_HW_QUEUES": "16"}), (8, 3, {"GPU_MAX_HW_QUEUES": "16"}),
                    (8, 3, {"GPU_MAX_HW_QUEUES": "16", "ZKPOA_UPLOAD_STREAM_PRIO": "high"}),
                    (8, 1, {"ZKPOA_UPLOAD_AFFINITY": "gpu"}), (12, 1, {}), (16, 1, {"GPU_MAX_HW_QUEUES": "16"})]
        for threads, streams, extra in variants:
            aff = " ".join("%s=%s" % kv for kv in extra.items())
            env = dict(os.environ, ZKPOA_UPLOAD_THREADS=str(threads), ZKPOA_UPLOAD_STREAMS=str(streams), **extra)
            rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--gb", str(args.gb), "--child", path], env=env,
                                capture_output=True, text=True)
            line = [l for l in rc.stdout.splitlines() if l.startswith("{")]
            print("threads %2d, own streams %d%s: %s" % (threads, streams, ", " + aff if aff else "",
                                                          line[-1] if line else "FAILED " + rc.stderr[-400:]), flush=True)
    finally:
        os.remove(path)


if __name__ == "__main__":
    main()
