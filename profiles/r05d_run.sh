# round 5, call d: price of an in-kernel BatchNorm statistics exchange (fused PixelCNN block question), allocation sites of a steady-state
# step, soak of the persistent launches with alternating exchange halves (batch sizes up to 128; and with the two-pass exact forward)
mkdir -p gpurun_out/r05d
timeout 120 profiles/microbench/bn_exchange_probe > gpurun_out/r05d/bn_exchange_probe.txt 2>&1; cat gpurun_out/r05d/bn_exchange_probe.txt
timeout 300 python profiles/microbench/alloc_probe.py > gpurun_out/r05d/alloc_probe.txt 2>&1; grep -v Warning gpurun_out/r05d/alloc_probe.txt | tail -25
timeout 600 python profiles/microbench/soak_persistent.py 3000 128 > gpurun_out/r05d/soak_b128.txt 2>&1; tail -2 gpurun_out/r05d/soak_b128.txt
timeout 600 python profiles/microbench/soak_persistent.py 1500 32 exact > gpurun_out/r05d/soak_exact.txt 2>&1; tail -2 gpurun_out/r05d/soak_exact.txt
