mkdir -p gpurun_out/r3n
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | cut -c1-600 | tail -30 > gpurun_out/r3n/gpu_tests.log
timeout 120 python tools/t_kernel_ms.py 2>&1 | tail -1 > gpurun_out/r3n/ms.log
timeout 600 python tools/t_est_soak.py 400 2>&1 | tail -15 > gpurun_out/r3n/soak.log
APX_LIB=/root/repo/apex_amd/lib/libapx_prof2.so timeout 300 python tools/t_prof.py 2>&1 | tail -60 > gpurun_out/r3n/prof2.log
APX_LIB=/root/repo/apex_amd/lib/libapx_prof.so timeout 300 python tools/t_prof.py 2>&1 | tail -30 > gpurun_out/r3n/prof.log
