timeout 600 python -m pytest tests/test_gpu_learner.py -q -x -m gpu 2>&1 | tail -3
timeout 300 python -m pytest tests -q -x -m gpu -k "recurrent or g15" 2>&1 | tail -3
for rep in 1 2; do
for v in "APX_X=0" "APX_X=1"; do env $v timeout 200 python bench.py --workload cassietraj_recurrent --steps 3 --warmup 1 --no_cpu_baseline 2>gpurun_out/rec.err | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', d['value'], d['sample_s'], d['optimize_s'])"; done
done
