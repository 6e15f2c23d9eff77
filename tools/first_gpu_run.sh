#!/bin/bash
# First GPU call of a round: the whole -m gpu suite on the final sources with the log kept (first line = kernel-source hash), then the three bench lines.
#     gpurun --timeout 3000 -- 'bash tools/first_gpu_run.sh r06'
set -u
TAG=${1:-r06}
OUT=gpurun_out/first_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import bench; print('kernel sources sha1:', bench.kernel_source_hash())" > "$OUT/gputest.log"
timeout 2400 python -m pytest tests -m gpu -q -rxXfE --durations=15 >> "$OUT/gputest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/gputest.log"
tail -40 "$OUT/gputest.log"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; tail -c 600 "$OUT/bench_line.json"
timeout 600 python bench.py --workload cassietraj_recurrent --steps 3 --warmup 1 > "$OUT/bench_line_recurrent.json" 2> "$OUT/bench_rec.err"; echo "rec rc=$?"; tail -c 400 "$OUT/bench_line_recurrent.json"
timeout 600 python bench.py --workload cassie_td3 --steps 5 --warmup 2 > "$OUT/bench_line_td3.json" 2> "$OUT/bench_td3.err"; echo "td3 rc=$?"; tail -c 300 "$OUT/bench_line_td3.json"
