# round 6, session e: the anatomy table again with the two extra what-if knobs (LDS staging of the gather, the granule loads themselves)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
python profiles/microbench/lstm_anatomy_probe.py 0 2 16 18 32 34 50 15 63 > $O/lstm_anatomy2.txt 2>&1
cat $O/lstm_anatomy2.txt
