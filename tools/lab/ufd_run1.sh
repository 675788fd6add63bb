mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -5
timeout 900 python tools/ufd_roll_ab.py > gpurun_out/ufd_ab_final.txt 2>&1; tail -2 gpurun_out/ufd_ab_final.txt | cut -c1-300
timeout 600 python tools/bench_ops.py > gpurun_out/bench_ops.txt 2>&1; grep -v "^\[" gpurun_out/bench_ops.txt | tail -16
