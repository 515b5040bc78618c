for ab in 0 1 2 3 4 7; do
echo "ablate $ab"; LELE_HIP_LAB=1 LELE_HIP_ATTN_ABLATE=$ab timeout 120 python tools/attention_bench.py --only default 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
print({k:[b['us'] for b in v.values()][0] for k,v in d.items()})"
done
