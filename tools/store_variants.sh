#!/bin/bash
# Build libcrec.so variants of the model kernels + rows_adam.hip with extra -D flags (crec_store.h's CREC_OUT_WT, rows_adam's
# CREC_ROWS_NT) and time each (GPU box).  usage: tools/store_variants.sh "-DCREC_OUT_WT=0" "-DCREC_OUT_WT=1 -DCREC_ROWS_NT=2" ...
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
FILES="ncf cml fm sbpr lrml transcf rows_adam"
n=0
for flags in "$@"; do
  n=$((n+1)); out=/tmp/crec_st_$n; mkdir -p $out
  pids=""
  for f in $FILES; do
    hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off $flags -c cleverrec_amd/csrc/$f.hip -o $out/$f.o & pids="$pids $!"
  done
  for p in $pids; do wait $p || exit 1; done
  objs=$(ls $O/*.o | grep -v -E "/(ncf|cml|fm|sbpr|lrml|transcf|rows_adam)\.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/*.o -L/opt/rocm/lib -lrccl || exit 1
  echo "== $flags"
  CREC_LIB=$out/libcrec.so python tools/prof_ncf.py 2>&1 | grep -E "^(NeuMF|GMF|FM|CML)"
  CREC_LIB=$out/libcrec.so python tools/bench_models.py 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-14}
done
