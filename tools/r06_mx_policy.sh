#!/bin/bash
# MXFP8 towers under the synchronizability head - agreement with the real reference and FT clips/s by operand policy (run on the GPU box).
#   SF_MX_BF16 = '' (every big Linear on MXFP8: the product mode) | fc2 | proj | proj,fc2 ; scale rule 0 (product) and 1 (tools/ab_build/libsf_mx1.so, if built)
cd ${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
out=gpurun_out/r06/mxpolicy; mkdir -p $out
ft() { python bench.py --workload ft --steps 10 --warmup 3 --no-cpu-baseline --no-workloads 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('FT', d['value'], 'clips/s', d['ms_per_step'], 'ms')"; }
for pol in "" fc2 proj proj,fc2; do
  echo "=== SF_MX_BF16='$pol' (scale rule 0)"
  SF_MX_BF16=$pol python tools/syncability_parity.py 2>&1 | grep mxfp8
  SF_MX_BF16=$pol python tools/fp8_trained_scale.py 2>&1 | grep "mxfp8 :"
  SF_MX_BF16=$pol ft
done > $out/policy.txt 2>&1
if [ -f tools/ab_build/libsf_mx1.so ]; then
  export SYNCHFORMER_HIP_LIB=$PWD/tools/ab_build/libsf_mx1.so
  for pol in "" proj,fc2; do
    echo "=== SF_MX_BF16='$pol' (scale rule 1)"
    SF_MX_BF16=$pol python tools/syncability_parity.py 2>&1 | grep mxfp8
    SF_MX_BF16=$pol python tools/fp8_trained_scale.py 2>&1 | grep "mxfp8 :"
  done >> $out/policy.txt 2>&1
fi
cat $out/policy.txt
