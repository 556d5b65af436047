#!/bin/bash
# A soak flag of the NCF family replayed: the shape under the rounding envelope (tools/check_ncf_shape.py, twenty draws) with the
# shipped library and with ncf.hip built with the given flags (e.g. the previous arithmetic).  GPU box.
# usage: tools/replay_ncf_flags.sh "<flags>" MLP 113 27 1000 2500 8 128 2 Adam
cd "$(dirname "$0")/.." || exit 1
flags="$1"; shift
O=cleverrec_amd/lib/obj; out=/tmp/crec_replay; mkdir -p $out
hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off $flags -c cleverrec_amd/csrc/ncf.hip -o $out/ncf.o || exit 1
objs=$(ls $O/*.o | grep -v "/ncf.o")
hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/ncf.o -L/opt/rocm/lib -lrccl || exit 1
echo "== shipped library"; python tools/check_ncf_shape.py "$@" 2>&1 | grep -E "seed|Error|envelope" | cut -c1-260
echo "== ncf.hip with $flags"; CREC_LIB=$out/libcrec.so python tools/check_ncf_shape.py "$@" 2>&1 | grep -E "seed|Error|envelope" | cut -c1-260
