#!/bin/bash
# First hardware run of apx_ppo_epoch (ppo_small.hip).  On the GPU box, from the repo root:
#     bash tools/epoch_gpu_check.sh [tag]          (through gpurun: gpurun --timeout 1500 -- 'bash tools/epoch_gpu_check.sh r05')
# 1. the GPU checks of the two persistent kernels (apx_ppo_epoch: golden G4b, twin of the per-step launches, PPO.update on / off; apx_td3_updates: golden G20b, twin of the
#    per-launch loop), each in its own process with a time limit;
# 2. the minibatch-64 bench line as launches and as one launch per epoch, the same for minibatch 256;
# 3. rocprofv3 kernel statistics of the one-launch run.
# Everything lands in gpurun_out/epoch_<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r05}
OUT=gpurun_out/epoch_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for mode in golden twin ppo td3_golden td3_twin; do
    timeout 600 python tests/epoch_worker.py $mode > "$OUT/check_$mode.jsonl" 2> "$OUT/check_$mode.err"
    echo "check $mode rc=$?" | tee -a "$OUT/summary.txt"
done
grep -h '"ok": false' "$OUT"/check_*.jsonl | head -20 | tee -a "$OUT/summary.txt"
for mb in 64 256; do
    timeout 900 python bench.py --steps 2 --warmup 1 --minibatch $mb --no_cpu_baseline > "$OUT/bench_mb${mb}_launches.json" 2> "$OUT/bench_mb${mb}_launches.err"
    timeout 900 python bench.py --steps 2 --warmup 1 --minibatch $mb --no_cpu_baseline --epoch_kernel > "$OUT/bench_mb${mb}_epoch.json" 2> "$OUT/bench_mb${mb}_epoch.err"
    for v in launches epoch; do
        python - "$OUT/bench_mb${mb}_$v.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", d["value"], "optimize_s", d["optimize_s"], "sample_s", d["sample_s"], "one_launch", d["config"].get("optimiser_steps_as_one_launch_per_epoch"))
except Exception as e:
    print(sys.argv[1], "no bench line:", e)
PY
    done
done
# TD3 (BASELINE configs[4]): the update block as launches and as one launch
for v in "" "--td3_one_launch"; do
    timeout 900 python bench.py --workload cassie_td3 --steps 3 --warmup 1 $v 2> /dev/null | tail -1 > "$OUT/bench_td3${v#--td3}.json"
    python -c "
import json, sys
try:
    d = json.loads(open('$OUT/bench_td3${v#--td3}.json').read().strip().splitlines()[-1]); print('td3 $v value', d['value'], 'ms_per_step', d['ms_per_step'], 'updates_per_s', d.get('updates_per_s'))
except Exception as e:
    print('td3 $v failed', e)" | tee -a "$OUT/summary.txt"
done
# the kernel against the reference's MuJoCo-generated tables (golden G24)
timeout 900 python tools/eval_ref_policy.py push 2>&1 | tail -4 | tee -a "$OUT/summary.txt"
timeout 900 python tools/eval_ref_policy.py commands 10000 2>&1 | tail -4 | tee -a "$OUT/summary.txt"
timeout 900 python tools/eval_ref_policy.py missions 2>&1 | tail -5 | tee -a "$OUT/summary.txt"
for wgs in 32 48 96 128; do      # grid size sweep at minibatch 64
    APX_PPO_EPOCH_WGS=$wgs timeout 600 python bench.py --steps 2 --warmup 1 --minibatch 64 --no_cpu_baseline --epoch_kernel 2> /dev/null | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wgs $wgs optimize_s', d['optimize_s'])
except Exception as e:
    print('wgs $wgs failed', e)" | tee -a "$OUT/summary.txt"
done
ROOT=$(pwd)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$ROOT/$OUT/prof" -- python "$ROOT/bench.py" --steps 2 --warmup 1 --minibatch 64 --no_cpu_baseline --epoch_kernel > "$ROOT/$OUT/prof.log" 2>&1 )
python tools/rocprof_summary.py "$(ls $OUT/prof/*/*.db | head -1)" "$OUT/kernel_stats_mb64_epoch.txt" > /dev/null 2>> "$OUT/summary.txt" || true
cat "$OUT/summary.txt"
