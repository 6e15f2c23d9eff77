cd /tmp && export TMPDIR=/tmp
for w in 256 512 1024 2048; do
  rm -rf /tmp/lt; APX_KSPLIT_WGS=$w rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python $GRAFT_REPO_ROOT/tools/t_learner_mb.py > /dev/null 2>&1
  echo "== target wgs $w"; python $GRAFT_REPO_ROOT/tools/t_learner_trace.py /tmp/lt | grep gemm | awk '{print $7}' | tr '\n' ' '; echo
done
