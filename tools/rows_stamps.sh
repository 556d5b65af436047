#!/bin/bash
# builds a -DCREC_ROWS_STAMPS variant of libcrec.so under /tmp and prints the per-wave timeline of CML's update launch (apply_all<32>) (GPU box)
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj; out=/tmp/crec_rows_stamps; mkdir -p $out
hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -Iinclude -DCREC_ROWS_STAMPS $@ -c cleverrec_amd/csrc/rows_adam.hip -o $out/rows_adam.o || exit 1
hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $(ls $O/*.o | grep -v "/rows_adam.o") $out/rows_adam.o -L/opt/rocm/lib -lrccl || exit 1
CREC_LIB=$out/libcrec.so python tools/rows_stamps.py
