#!/bin/bash
# Build libcrec.so variants of bpr_train.hip with extra -D flags and time a 20-batch tf1 call with each (GPU box).
# usage: tools/idx_variants.sh "-DCREC_IDX_RECS=4" "-DCREC_IDX_RECS=12" ...
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
n=0
for flags in "$@"; do
  n=$((n+1)); out=/tmp/crec_idx_$n; mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude $flags -c cleverrec_amd/csrc/bpr_train.hip -o $out/bpr_train.o || exit 1
  objs=$(ls $O/*.o | grep -v "/bpr_train.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/bpr_train.o -L/opt/rocm/lib -lrccl || exit 1
  echo "== $flags"; CREC_LIB=$out/libcrec.so python tools/prof_call.py 2>&1 | grep "C call" | tail -1
done
