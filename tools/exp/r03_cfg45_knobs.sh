set -o pipefail
B="--no-cpu-baseline --no-open-compare --no-two-streams --no-other-configs --no-host-api --steps 20 --warmup 3"
run() { python bench.py $B "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f  step %.4f ms  %s  verified=%s' % (d['value'], d['ms_per_step'], d['kernels_avg_ms'], d['verified_vs_oracle']))"; }
for c in 4 5; do echo "config $c baseline"; run --config $c; for v in 16n 8n 4n; do echo "config $c KMX_FILL_VARIANT=$v"; KMX_FILL_VARIANT=$v run --config $c; done; done
echo "== KMX_LOOKUP_ITEMS=8 build"
KMX_LOOKUP_ITEMS=8 python -c "from kmer_index_amd import build; build.build(force=True)"
for c in 2 4 5; do echo "config $c items=8"; run --config $c; done
echo "== KMX_LOOKUP_ITEMS=2 build"
KMX_LOOKUP_ITEMS=2 python -c "from kmer_index_amd import build; build.build(force=True)"
for c in 2 4 5; do echo "config $c items=2"; run --config $c; done
