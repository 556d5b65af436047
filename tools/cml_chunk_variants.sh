#!/bin/bash
# CML: slots per chunk of the per-step item index (GPU box).  usage: tools/cml_chunk_variants.sh 2048 4096 6144 12288
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
for cs in "$@"; do
  out=/tmp/crec_cmlc_$cs; mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DCML_CHUNK_SLOTS=$cs -c cleverrec_amd/csrc/cml.hip -o $out/cml.o || exit 1
  objs=$(ls $O/*.o | grep -v "/cml.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/cml.o -L/opt/rocm/lib -lrccl || exit 1
  echo "== $cs slots per chunk"; CREC_LIB=$out/libcrec.so python tools/prof_ncf.py 2>&1 | grep "^CML"
done
