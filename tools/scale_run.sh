#!/bin/bash
# The driver's scaling run, reproduced: N = 1, 2, 4, 8 ranks of bench.py on ONE node, one rank per GPU over RCCL, back to back, then the efficiency table
# the driver computes from the four JSON lines.  Usage (on an 8-GPU MI355X node, repo root):  bash tools/scale_run.sh [steps] [warmup]
#   APX_BENCH_SHARE_GPU=1 bash tools/scale_run.sh 2 1   -> the same command lines with every rank on cuda:0 over gloo (control-flow check on a 1-GPU box)
set -e
STEPS=${1:-5}; WARMUP=${2:-1}
OUT=gpurun_out/scale; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=29517
for N in 1 2 4 8; do
  if [ "$N" = "1" ]; then
    python bench.py --gpus 1 --steps $STEPS --warmup $WARMUP --no_cpu_baseline 2>$OUT/n$N.err | tail -1 > $OUT/n$N.json
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + N)) bench.py --gpus $N --steps $STEPS --warmup $WARMUP 2>$OUT/n$N.err | grep '^{' | tail -1 > $OUT/n$N.json
  fi
  echo "N=$N: $(cut -c1-160 $OUT/n$N.json)"
done
# the recurrent workload (BASELINE.json configs[3]) over the same N
for N in 1 2 4 8; do
  if [ "$N" = "1" ]; then
    python bench.py --workload cassietraj_recurrent --gpus 1 --steps 2 --warmup 1 --no_cpu_baseline 2>$OUT/rec_n$N.err | tail -1 > $OUT/rec_n$N.json
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((PORT + 10 + N)) bench.py --workload cassietraj_recurrent --gpus $N --steps 2 --warmup 1 2>$OUT/rec_n$N.err | grep '^{' | tail -1 > $OUT/rec_n$N.json
  fi
  echo "recurrent N=$N: $(cut -c1-160 $OUT/rec_n$N.json)"
done
python - <<'PY'
import json
rows = {n: json.load(open("gpurun_out/scale/n%d.json" % n)) for n in (1, 2, 4, 8)}
base = rows[1]["value"]
print("%4s %16s %12s %12s %22s" % ("N", "env-steps/s", "ms/step", "efficiency", "all-reduce ms / step"))
for n, r in rows.items():
    c = r.get("collectives", {})
    print("%4d %16.0f %12.2f %12.3f %22s" % (n, r["value"], r["ms_per_step"], r["value"] / (n * base), c.get("allreduce_ms_per_step")))
    pr = c.get("per_rank") or {}
    for k in ("sample_s", "optimize_s", "allreduce_ms_per_step"):      # what every rank saw: a straggler rank or an exposed collective shows here, not in rank 0's line
        if k in pr: print("       per-rank %-22s %s" % (k, " ".join("%.4g" % x for x in pr[k])))
json.dump({"per_n": {n: {"value": r["value"], "ms_per_step": r["ms_per_step"], "efficiency": r["value"] / (n * base), "collectives": r.get("collectives")} for n, r in rows.items()}},
          open("gpurun_out/scale/summary.json", "w"), indent=1)
PY
