#!/bin/bash
# SQ counters of the persistent 3 x 3 window kernel on 64 -> 64 channels at 160 x 160 x 64 images
bash tools/pmc_kernel.sh conv_window_p_kernel r5g_pmc -- python $GRAFT_REPO_ROOT/tools/conv_ab.py --only "64->64 k3 s1 @160" --iters 3 2>&1 | tail -30
