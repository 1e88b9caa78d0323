# k_validate_walk (linear intersection) against k_validate<false> (per-candidate search, KMX_VALIDATE_SEARCH=1): config 3 and
# STITCH sweeps
B="--no-cpu-baseline --no-open-compare --no-two-streams --no-other-configs --no-host-api --steps 20 --warmup 3"
run() { python bench.py $B "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f  step %.4f ms  %s  verified=%s' % (d['value'], d['ms_per_step'], d['kernels_avg_ms'], d['verified_vs_oracle']))"; }
echo "config 3 walk"; run --config 3
echo "config 3 search"; KMX_VALIDATE_SEARCH=1 run --config 3
export KMX_SWEEP_N=100000000 KMX_SWEEP_LENS=11,13,20,23,31,64
echo "sweep walk"; python tools/probe_sweep.py "dna4 k=10" 2>/dev/null
echo "sweep search"; KMX_VALIDATE_SEARCH=1 python tools/probe_sweep.py "dna4 k=10" 2>/dev/null
