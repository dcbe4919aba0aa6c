R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags $R/synchformer_amd/csrc/*.hip -o /tmp/libsf_ab$i.so || exit 1
done
for rep in 1 2 3; do
  i=0
  for flags in "$@"; do
    i=$((i+1))
    SYNCHFORMER_HIP_LIB=/tmp/libsf_ab$i.so python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $i [$flags]', d['value'], d['ms_per_step'])"
  done
done
