#!/bin/bash
# builds a -DCREC_STAMPS variant of libcrec.so under /tmp and prints the per-wave timeline of bpr_step_fused (GPU box)
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj; out=/tmp/crec_stamps; mkdir -p $out
hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DCREC_STAMPS $@ -c cleverrec_amd/csrc/bpr_train.hip -o $out/bpr_train.o || exit 1
hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $(ls $O/*.o | grep -v "/bpr_train.o") $out/bpr_train.o -L/opt/rocm/lib -lrccl || exit 1
CREC_LIB=$out/libcrec.so python ${STAMP_TOOL:-tools/wave_stamps.py}
