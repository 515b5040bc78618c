#!/usr/bin/env python3
"""Phase timing of attention_kernel from inside the kernel (developer switch LELE_HIP_ATTN_STAMPS = device address of a
[workgroups][8] i64 buffer): thread 0 of every workgroup stamps the shader clock at the phase boundaries.  Prints the median
cycles per phase over the workgroups (cycle counters are per XCD: only differences inside a workgroup are used)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H, DH = 4, 128
QC = [["slice", 2, 0, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]
KC = [["slice", 2, 512, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 3, 1]]]
VC = [["slice", 2, 1024, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]


def main():
    import torch
    from lele_amd import kernels as K
    from lele_amd._lib import Ctx, Weight
    ctx = Ctx()
    rng = np.random.default_rng(0)
    scale = Weight(np.array([DH ** -0.5], np.float32))
    names = ["first loads + Q", "scores (own tiles)", "barrier", "softmax (own rows)", "barrier", "P V products", "stores"]
    for b, t in ((32, 171), (1, 504)):
        qd = ctx.buf().upload((rng.standard_normal((b, t, 1536)) * 1.5).astype(np.float32))
        dst = ctx.buf()
        os.environ["LELE_HIP_ATTENTION_MIN_BLOCKS"] = "1"
        call = lambda: K.attention_view(qd, QC, qd, KC, qd, VC, scale, [0, 2, 1, 3], [0, 0, H * DH], out=dst, ctx=ctx)
        for _ in range(3):
            call()
        ctx.sync()
        rows = int(os.environ.get("LELE_HIP_ATTENTION_ROWS", "0")) or (16 if b * H * ((t + 31) // 32) < 128 else 32)
        nwg = b * H * ((t + rows - 1) // rows)
        dbg = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        os.environ["LELE_HIP_ATTN_STAMPS"] = hex(dbg.data_ptr())
        call()
        ctx.sync()
        del os.environ["LELE_HIP_ATTN_STAMPS"]
        tt = dbg.cpu().numpy().astype(np.float64)
        d = np.diff(tt[:, :7], axis=1)
        ctx.timer_start()
        for _ in range(20):
            call()
        us = ctx.timer_stop() * 1e3 / 20
        print("%d x %d rows: %.1f us per call; median cycles per phase [max]:" % (b, t, us))
        # the stamp layout is start, [1]..[6]
        labels = ["scores (own tiles, incl. first loads)", "barrier", "softmax (own rows)", "barrier", "P V products", "stores"]
        for i, nm in enumerate(labels):
            print("   %-40s %7.0f [%7.0f]" % (nm, np.median(d[:, i]), d[:, i].max()))
        print("   whole life median %.0f" % np.median(tt[:, 6] - tt[:, 0]))


if __name__ == "__main__":
    main()
