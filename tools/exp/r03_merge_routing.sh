# m = 9 / m = 8 without prefix levels: the register paths of k_prefix_sort_small (<= 4 runs, <= 512 positions) against the
# merge-path kernel for the same slices (KMX_PMERGE_REG_RUNS=0 routes them to k_prefix_merge_small)
export KMX_SWEEP_N=100000000 KMX_PREFIX_LEVELS=-1 KMX_SWEEP_LENS=8,9
echo "== default routing"; python tools/probe_sweep.py "dna4 k=10" 2>/dev/null
echo "== KMX_PMERGE_REG_RUNS=0"
KMX_PMERGE_REG_RUNS=0 python -c "from kmer_index_amd import build; build.build(force=True)"
python tools/probe_sweep.py "dna4 k=10" 2>/dev/null
KMX_SWEEP_LENS=4 python tools/probe_sweep.py "aa20 k=5" 2>/dev/null
