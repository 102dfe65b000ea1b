# round 6, session b: full GPU suite on the new tree, the default bench line (vendor-stack yardstick included), and the
# four-ranks-on-one-GPU question (which variable makes --gpus 4 slow on a shared device: size, wire format, rank count)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
( time timeout 1200 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
for v in "--workload toy --dtype f32" "--dp-payload f32" "--dp-mode encoder_only"; do
  n=$(echo $v | tr -d ' -' )
  ( time timeout 200 python bench.py --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline --launch-timeout 150 $v ) > $O/g4_$n.json 2> $O/g4_$n.err
  echo "== $v rc=$?"; head -c 400 $O/g4_$n.json; grep -E "real|attempt" $O/g4_$n.err | tail -3
done
( time timeout 200 python bench.py --gpus 3 --steps 3 --warmup 1 --no-cpu-baseline --launch-timeout 150 ) > $O/g3.json 2> $O/g3.err
echo "== gpus 3 rc=$?"; head -c 400 $O/g3.json; grep -E "real|attempt" $O/g3.err | tail -3
