# A/B of kernel variants: `bash tools/ab_variants.sh out_dir lib1 lib2 ...` (library base names under apex_amd/lib): quick parity tests + kernel ms for each
out=gpurun_out/$1; shift; mkdir -p $out
for L in "$@"; do
  echo "== $L"
  APX_LIB=/root/repo/apex_amd/lib/$L.so timeout 300 python -m pytest tests/test_gpu_env.py -q -x -k "substeps_track_oracle or env_steps_vs_oracle or single_substep_crafted or saturation_flags_vs_oracle_crafted or invariants" 2>&1 | tail -3
  APX_LIB=/root/repo/apex_amd/lib/$L.so timeout 120 python tools/t_kernel_ms.py 2>&1 | tail -1
  APX_LIB=/root/repo/apex_amd/lib/$L.so timeout 120 python tools/t_kernel_ms.py 2>&1 | tail -1
done > $out/ab.log 2>&1
cat $out/ab.log
