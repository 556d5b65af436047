#!/bin/bash
# Build libcrec.so variants of ncf.hip with extra -D flags and profile each (GPU box).
# usage: tools/ncf_variants.sh "-DNCF_NT=256" "-DNCF_NT=512 -DNCF_STAMPS" ...
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
n=0
for flags in "$@"; do
  n=$((n+1)); out=/tmp/crec_ncf_$n; mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off $flags -c cleverrec_amd/csrc/ncf.hip -o $out/ncf.o || exit 1
  objs=$(ls $O/*.o | grep -v "/ncf.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/ncf.o -L/opt/rocm/lib -lrccl || exit 1
  echo "== $flags"; CREC_LIB=$out/libcrec.so python tools/prof_ncf.py 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-40}
done
