cd /tmp && export TMPDIR=/tmp
for a in 0 1 2 3; do
rm -rf /tmp/yp; C3ABL=$a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/yp -o n64 -- python $GRAFT_REPO_ROOT/tools/yolo_graph.py --batch 64 --check 0 --runs 5 > /dev/null 2>&1; echo "abl $a: $(grep conv3x3_mfma /tmp/yp/n64_kernel_stats.csv | cut -c160-260)"
done
