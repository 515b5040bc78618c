#!/usr/bin/env python3
"""Phase timing of the attention kernels from inside (lab build only: LELE_HIP_LAB=1; lab switch LELE_HIP_ATTN_STAMPS = device
address of an i64 buffer): the shader clock is stamped at the phase boundaries.  Prints median cycles over the workgroups (cycle
counters are per XCD: only differences inside a workgroup are used) -- the batch kernel's per-tile timeline for 32 x 171 rows,
the 16-row kernel's phases for 1 x 504."""
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
        if b * H * ((t + 127) // 128) >= 128 and "LELE_HIP_ATTENTION_ROWS" not in os.environ:
            # the batch kernel (attention_flash_kernel): [workgroups][8 waves][64] stamps.  Compute wave: 0 start, 1 Q split, 2 first
            # barrier, then per key tile 3 + 4 i: scores out of the matrix core / softmax done / P V issued / barrier passed; 63 stores
            # issued.  Producer wave: 0 start, 1 tile 0 requested, 2 tile 0 split + tiles 1, 2 requested, 3 first barrier, then per tile
            # 4 + 3 i: next fetch issued / next tile split and stored / barrier passed.
            nwg = b * H * ((t + 127) // 128)
            dbg = torch.zeros((nwg, 8, 64), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            os.environ["LELE_HIP_ATTN_STAMPS"] = hex(dbg.data_ptr())
            call()
            ctx.sync()
            del os.environ["LELE_HIP_ATTN_STAMPS"]
            tt = dbg.cpu().numpy().astype(np.float64)
            t0 = np.where(tt[:, :, 0] > 0, tt[:, :, 0], np.inf).min(axis=1, keepdims=True)
            full = [wg for wg in range(nwg) if tt[wg, 3, 3] > 0]   # workgroups whose four compute waves all have rows
            nt = (t + 31) // 32
            ctx.timer_start()
            for _ in range(20):
                call()
            us = ctx.timer_stop() * 1e3 / 20
            print("%d x %d rows, batch kernel: %.1f us per call (stamped build); median cycles since the workgroup's first stamp, %d full workgroups" % (b, t, us, len(full)))
            c = tt[full][:, 0] - t0[full]
            print("   compute wave 0:  Q split %d | first barrier %d" % (np.median(c[:, 1]), np.median(c[:, 2])))
            for i in range(nt):
                m = np.median(c[:, 3 + 4 * i:7 + 4 * i], axis=0)
                print("      tile %d: scores %6d  softmax %6d  P V issued %6d  barrier %6d" % (i, *m))
            print("      stores issued %d" % np.median(c[:, 63]))
            for w, nm in ((4, "K producer"), (6, "V producer")):
                pr = tt[full][:, w] - t0[full]
                print("   %s (wave %d): tile 0 requested %d | split + stored %d | first barrier %d" % (nm, w, *np.median(pr[:, 1:4], axis=0)))
                for i in range(nt):
                    m = np.median(pr[:, 4 + 3 * i:7 + 3 * i], axis=0)
                    print("      tile %d: fetch issued %6d  next tile split %6d  barrier %6d" % (i, *m))
            continue
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
