# round 6, session d: (1) anatomy of a timestep of the persistent recurrences by what-if builds of the FINAL kernels (VERDICT r5
# item 4; the LV_TRACE build of session c runs 35 % slower per timestep than the product even untraced, its phase split is kept as a
# second view); (2) four ranks on ONE GPU with the host-staged gloo exchange; (3) the default line with the vendor-stack yardstick
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
python profiles/microbench/lstm_anatomy_probe.py > $O/lstm_anatomy.txt 2>&1
cat $O/lstm_anatomy.txt
for n in 2 4 8; do
  ( time timeout 400 python bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --launch-timeout 300 ) > $O/g$n.json 2> $O/g$n.err
  echo "== gpus $n rc=$?"; head -c 300 $O/g$n.json; echo; grep -E "^real" $O/g$n.err
done
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
for l in open("gpurun_out/r06d/bench_default.json"):
    if l.startswith("{"):
        d = json.loads(l)
        print(d["value"], d["ms_per_step"], json.dumps(d.get("vendor_stack_baseline"))[:1500])
PY
grep -E "^real" $O/bench_default.err
