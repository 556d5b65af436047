#!/bin/bash
# Like ncf_variants.sh, and runs tests/test_gpu_ncf.py against every variant library (GPU box).
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
n=0
for flags in "$@"; do
  n=$((n+1)); out=/tmp/crec_ncf_$n; mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off $flags -c cleverrec_amd/csrc/ncf.hip -o $out/ncf.o || exit 1
  objs=$(ls $O/*.o | grep -v "/ncf.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/ncf.o -L/opt/rocm/lib -lrccl || exit 1
  echo "== $flags"; CREC_LIB=$out/libcrec.so python tools/prof_ncf.py 2>&1 | grep -E "^(NeuMF|GMF)"
  CREC_LIB=$out/libcrec.so python -m pytest tests/test_gpu_ncf.py -m gpu -x -q 2>&1 | tail -2
done
