cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "0.02,0.1,2.0" "0.01,0.05,4.0"; do echo "DAMP $d"; LELE_SV_DAMP=$d python -m pytest tests/test_graph_oracle.py -m gpu -x -q -k sensevoice -s > /dev/null 2>&1; python -c "
import json
d=json.load(open('gpurun_out/sensevoice_graph_oracle.json'))
for k,v in d.items():
    if k.startswith('configs'): print(k, round(v['argmax_agreement'],4), round(v['mae_over_rms'],5), round(v['worst_gap_at_a_differing_frame_over_rms'],4))
"; done
