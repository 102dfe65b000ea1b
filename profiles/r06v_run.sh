#!/bin/bash
# round 6, call v: does the abort of call u's GPU suite (inside test_image_step_split_bf16_convolutions_hold_the_f32_bounds, during a Python GC) repeat?
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
for i in 1 2; do
  python -m pytest tests -q -m gpu > $O/r06v_pytest_gpu_$i.txt 2>&1; echo "full suite run $i: rc=$?"; tail -2 $O/r06v_pytest_gpu_$i.txt | cut -c1-200
done
for i in 1 2 3; do
  python -m pytest tests/test_gpu_parity.py -q -m gpu -k "image" > $O/r06v_pytest_image_$i.txt 2>&1; echo "image tests run $i: rc=$?"; tail -1 $O/r06v_pytest_image_$i.txt | cut -c1-200
done
