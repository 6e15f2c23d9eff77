mkdir -p gpurun_out/r3v
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | cut -c1-600 | tail -12 > gpurun_out/r3v/gpu_tests.log
timeout 120 python tools/t_kernel_ms.py 2>&1 | tail -1 > gpurun_out/r3v/ms.log
APX_LIB=/root/repo/apex_amd/lib/libapx_prof.so timeout 300 python tools/t_prof.py 2>&1 | tail -12 > gpurun_out/r3v/prof.log
APX_LIB=/root/repo/apex_amd/lib/libapx_prof2.so timeout 300 python tools/t_prof.py 2>&1 | tail -30 > gpurun_out/r3v/prof2.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r3v/smoke.log
