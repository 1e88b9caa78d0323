for v in KMX_EXP_NO_GATOMIC KMX_EXP_NO_STORE; do
  echo "== $v"
  env $v=1 python -c "from kmer_index_amd import build; build.build(force=True)" || exit 1
  timeout -k 10 200 python tools/probe_prefix.py 3 2>&1 | grep "^m="
done
python -c "from kmer_index_amd import build; build.build(force=True)"
