# round 6, session h: Omniglot with the shortened BatchNorm statistics prologue (bn_block_totals: 32 partial rows per round trip, second
# level read-then-add): image parity tests, throughput, kernel trace
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "image or bn or omniglot or pixelcnn or conv" > $O/pytest_image.txt 2>&1; tail -3 $O/pytest_image.txt
for d in bf16x3 f32; do
  python bench.py --workload omniglot --dtype $d --graph 1 --steps 30 --warmup 5 --no-cpu-baseline --no-vendor-baseline > $O/omniglot_${d}_hipgraph.json 2>/dev/null
  cut -c1-220 $O/omniglot_${d}_hipgraph.json
done
python profiles/microbench/omniglot_bn_floor.py bf16x3 > $O/omniglot_bn_floor_bf16x3.txt 2>&1
tail -5 $O/omniglot_bn_floor_bf16x3.txt | cut -c1-200
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_o -o o -- python $GRAFT_REPO_ROOT/bench.py --workload omniglot --dtype bf16x3 --steps 10 --warmup 2 --no-cpu-baseline --no-vendor-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocpd.py $O/prof_o/o_results.db > $O/omniglot_kernel_stats.txt 2>&1
head -16 $O/omniglot_kernel_stats.txt | cut -c1-70,112-175
rm -rf $O/prof_o
