#!/bin/bash
# Registers / scratch / occupancy of every kernel in one source file (measurement tooling; runs where hipcc is, no GPU needed):
#   bash profiles/kernel_resources.sh lv_gemm_b16.hip [name filter]
# A non-zero ScratchSize on a hot kernel is a spill (round 4: wrapping the 256-tile kernel's body in a loop over tiles cost its
# plain-store variants ~1 KB of scratch per lane and 60 % of their speed; the build that shipped has 0 / 32 bytes).
cd "$(dirname "$0")/.."
C=vae_lagging_encoder_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I $C -c $C/$1 -o /tmp/kr.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cur, vals = None, {}
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln) or re.search(r' Name: (\S+)', ln)
    if m:
        cur, vals = m.group(1), {}
    for k in ('TotalSGPRs:', ' VGPRs:', 'AGPRs:', 'VGPRs Spill:', 'ScratchSize [bytes/lane]:', 'Occupancy [waves/SIMD]:', 'LDS Size [bytes/block]:'):
        m = re.search(re.escape(k) + r' (\d+)', ln)
        if m and cur:
            vals[k.strip(' :').split(' [')[0]] = int(m.group(1))
    if 'LDS Size' in ln and cur and flt in cur:
        print('%-90s %s' % (cur[:90], vals))
" "${2:-}"
