mkdir -p gpurun_out/r04e
M=profiles/microbench
timeout 300 python $M/gemm_lstm_shapes.py > gpurun_out/r04e/gemm_lstm_shapes2.txt 2>&1; cat gpurun_out/r04e/gemm_lstm_shapes2.txt
