#!/bin/bash
# B fragments a tap ahead in the 32-channel-block form of the 3 x 3 window kernel (new) against reading them where they are used (prev)
timeout 300 python tools/conv_ab.py --only "k3 s1" --out gpurun_out/pf_new.json > /dev/null 2>&1
LELE_HIP_LIBRARY=liblele_hip_prev.so timeout 300 python tools/conv_ab.py --only "k3 s1" --out gpurun_out/pf_prev.json > /dev/null 2>&1
python tools/conv_ab.py --compare gpurun_out/pf_prev.json gpurun_out/pf_new.json
timeout 600 python -m pytest tests/test_conv_rnn.py -m gpu -x -q 2>&1 | tail -2
