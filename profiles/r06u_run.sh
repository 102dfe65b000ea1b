#!/bin/bash
# round 6, call u: final GPU suite, smoke(), and the soaks (varying shapes; plain, with the exchange forced on over one-rank RCCL, B <= 128) on the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
python -m pytest tests -q -m gpu > $O/r06u_pytest_gpu.txt 2>&1; tail -3 $O/r06u_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r06u_smoke.txt 2>&1; tail -2 $O/r06u_smoke.txt
timeout 900 python profiles/microbench/soak_persistent.py 6000 32 > $O/r06u_soak_plain.txt 2>&1; tail -2 $O/r06u_soak_plain.txt | cut -c1-250
timeout 600 python profiles/microbench/soak_persistent.py 3000 32 dp > $O/r06u_soak_dp.txt 2>&1; tail -3 $O/r06u_soak_dp.txt | cut -c1-250
timeout 600 python profiles/microbench/soak_persistent.py 1500 128 dp > $O/r06u_soak_dp_b128.txt 2>&1; tail -3 $O/r06u_soak_dp_b128.txt | cut -c1-250
