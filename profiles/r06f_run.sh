# round 6, session f: (1) knob sweep of the 16-row persistent kernels (LV_HB16 / LV_SBB16 on the final kernel, plus the
# forward's LV_SBK16 / LV_GJ16); (2) which memory-side counters this rocprofv3 offers (MALL / DRAM split of the L2's fabric requests)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
python profiles/microbench/lstm_anatomy_probe.py 0 sbb4 sbb2 sbb16 hb4sbb4 hb1 sbk2 sbk8 gj8 gj32 > $O/persist16_knob_sweep.txt 2>&1
cat $O/persist16_knob_sweep.txt | cut -c1-130
export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_avail.txt 2>&1
grep -i -E "mall|dram|EA0_RDREQ|EA0_WRREQ|TCC_HIT|TCC_MISS|TCC_REQ" $GRAFT_REPO_ROOT/$O/counters_avail.txt | cut -c1-220 | head -40
