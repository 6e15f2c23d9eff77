mkdir -p gpurun_out/r3q
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | cut -c1-600 | tail -12 > gpurun_out/r3q/gpu_tests.log
timeout 600 python bench.py 2> gpurun_out/r3q/bench.err | tail -1 > gpurun_out/r3q/bench.json
