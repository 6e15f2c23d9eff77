# recurrent bench line under two builds, alternating: bash tools/ab_rec.sh libA.so libB.so
A=${1:-apex_amd/lib/libapx.so}; B=${2:-$A}
for rep in 1 2; do for L in $A $B; do
APX_LIB=$PWD/$L timeout 200 python bench.py --workload cassietraj_recurrent --steps 3 --warmup 1 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', 'rec', d['value'], d['sample_s'], d['optimize_s'])"
done; done
