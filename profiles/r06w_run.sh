#!/bin/bash
# round 6, call w: bisecting the abort of the GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
LVAE_PAIR_WGRAD=0 python -m pytest tests -q -m gpu > $O/r06w_nopair_engine.txt 2>&1; echo "engine without the pair launch (kernel test still runs): rc=$?"; tail -1 $O/r06w_nopair_engine.txt | cut -c1-200
python -m pytest tests -q -m gpu -k "not gemm_b16_pair" > $O/r06w_nopair_test.txt 2>&1; echo "without the pair kernel test: rc=$?"; tail -1 $O/r06w_nopair_test.txt | cut -c1-200
LVAE_PAIR_WGRAD=0 python -m pytest tests -q -m gpu -k "not gemm_b16_pair" > $O/r06w_nopair_at_all.txt 2>&1; echo "without either: rc=$?"; tail -1 $O/r06w_nopair_at_all.txt | cut -c1-200
