for rep in 1 2 3; do for L in "$@"; do printf "%s " $L; APX_LIB=/root/repo/apex_amd/lib/$L.so timeout 120 python tools/t_kernel_ms.py 2>&1 | tail -1; done; done
