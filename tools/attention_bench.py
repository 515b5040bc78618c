#!/usr/bin/env python3
"""Time lele_hip_attention_view at the SenseVoice-shaped sizes (BASELINE configs[2]: 1 x 504 rows, configs[3]: 32 x 171 rows;
4 heads of 128) as the library dispatches it, as the f32 replica, and as the three-call sequence it replaces (with the lab
library also the row-block forms selected by hand).
Twenty calls are captured into a hipGraph and its replays timed with HIP events.  Output: one JSON object."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

H, DH = 4, 128
QC = [["slice", 2, 0, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]
KC = [["slice", 2, 512, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 3, 1]]]
VC = [["slice", 2, 1024, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    from lele_amd import kernels as K
    from lele_amd._lib import Ctx, Weight
    ctx = Ctx()
    rng = np.random.default_rng(0)
    scale = Weight(np.array([DH ** -0.5], np.float32))
    out = {}
    for name, (b, t) in (("c4 32x171", (32, 171)), ("c3 1x504", (1, 504)), ("8x171", (8, 171)), ("64x171", (64, 171))):
        qd = ctx.buf().upload((rng.standard_normal((b, t, 1536)) * 1.5).astype(np.float32))
        dst = ctx.buf()
        flops = 2 * 2 * b * H * t * t * DH
        row = {}
        variants = [("default (one pass over the keys for a batch)", {"LELE_HIP_ATTENTION_MIN_BLOCKS": "1"}),
                    ("replica (LELE_HIP_ATTENTION_EXACT=1)", {"LELE_HIP_ATTENTION_MIN_BLOCKS": "1", "LELE_HIP_ATTENTION_EXACT": "1"}),
                    ("sequence", {"LELE_HIP_ATTENTION_FUSED": "0"})]
        if os.environ.get("LELE_HIP_LAB") == "1":  # the row-block forms by hand: lab-only switches
            variants += [("fused rt=1", {"LELE_HIP_ATTENTION_RT": "1", "LELE_HIP_ATTENTION_MIN_BLOCKS": "1", "LELE_HIP_ATTENTION_ROWS": "32"}),
                         ("fused rt=2", {"LELE_HIP_ATTENTION_RT": "2", "LELE_HIP_ATTENTION_MIN_BLOCKS": "1", "LELE_HIP_ATTENTION_ROWS": "32"}),
                         ("fused 16 rows", {"LELE_HIP_ATTENTION_RT": "1", "LELE_HIP_ATTENTION_MIN_BLOCKS": "1", "LELE_HIP_ATTENTION_ROWS": "16"})]
        for label, env in variants:
            if args.only and args.only not in label:
                continue
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                call = lambda: K.attention_view(qd, QC, qd, KC, qd, VC, scale, [0, 2, 1, 3], [0, 0, H * DH], out=dst, ctx=ctx)
                call()
                ctx.sync()
                ctx.graph_begin()
                for _ in range(20):
                    call()
                gr = ctx.graph_end()
                gr.launch()
                ctx.sync()
                ctx.timer_start()
                for _ in range(args.reps // 20):
                    gr.launch()
                ms = ctx.timer_stop() / (args.reps // 20 * 20)
                gr.close()
            finally:
                for k, v in old.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            row[label] = {"us": round(ms * 1e3, 2), "tflops_f32": round(flops / ms / 1e9, 2)}
        out[name] = row
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
