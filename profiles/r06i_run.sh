# round 6, session i: A/B of the BatchNorm statistics prologue (bn_block_totals): rows beyond nblk re-read row 0 (product: every thread of
# every workgroup hits the same line) or the thread's own first row (LV_BN_TOT_CLAMP_OWN), 8 / 16 / 32 rows per round trip
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
python profiles/microbench/omniglot_ab.py product l2u > $O/omniglot_ab6.txt 2>&1
cat $O/omniglot_ab6.txt
