"""The section-8(e) recogniser loop on the device (tools/sensevoice_sharded.py): front-end, CMVN, compiled encoder as a hipGraph,
decode on the device and the all-gather of token ids over RCCL -- here with a 1-rank group (one GPU per test box) and a
two-layer encoder; the world_size-2 logic of the gather runs on CPU over gloo in test_multiprocess.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_sharded_recogniser_one_rank_rccl(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = tmp_path / "rec.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sensevoice_sharded.py"), "--dist", "--layers", "2", "--per-gpu", "4",
                        "--seconds", "3", "--steps", "2", "--warmup", "1", "--out", str(out)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(out.read_text())
    assert rec["ranks_agree"] is True and rec["n_gpus"] == 1 and rec["utterances"] == 4
    assert rec["collective"] == "rccl all-gather of token ids" and rec["gathered_bytes_per_gpu"] == 4 * (1 + 50 + 4) * 4   # 50 LFR rows + 4 prompt tokens, count-prefixed
    assert 0 < rec["value"] < 1e-2 and rec["ms_per_step"] > 0
