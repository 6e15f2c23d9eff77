# headline bench line of the library under APX_LIB, alternating between two builds: bash tools/ab_main.sh libA.so libB.so
A=${1:-apex_amd/lib/libapx.so}; B=${2:-$A}
for rep in 1 2; do for L in $A $B; do
APX_LIB=$PWD/$L timeout 300 python bench.py --steps 10 --warmup 2 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$L', 'ppo', d['value'], d['sample_s'], d['optimize_s'], d['roofline'].get('mlp_forward_mfma'))"
done; done
