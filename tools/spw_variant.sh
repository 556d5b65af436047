#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj; out=/tmp/crec_spw32; mkdir -p $out
hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DCREC_LAZY_SPW=32 -c cleverrec_amd/csrc/bpr_train.hip -o $out/bpr_train.o || exit 1
hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $(ls $O/*.o | grep -v "/bpr_train.o") $out/bpr_train.o -L/opt/rocm/lib -lrccl || exit 1
CREC_LIB=$out/libcrec.so WS="0 8" python tools/time_window.py
