#!/bin/bash
# S-big under lazy_window: build libcrec.so variants of bpr_train.hip (one per argument = its -D flags) and time them (GPU box).
# usage: W=8 tools/window_variants.sh "-DCREC_WIN_DEEP=8" "-DCREC_WIN_DEEP=12" ...
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
for defs in "$@"; do
  out=/tmp/crec_win_$(echo $defs | tr -c 'A-Za-z0-9' _); mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude $defs -c cleverrec_amd/csrc/bpr_train.hip -o $out/bpr_train.o || exit 1
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $(ls $O/*.o | grep -v "/bpr_train.o") $out/bpr_train.o -L/opt/rocm/lib -lrccl || exit 1
  CREC_LIB=$out/libcrec.so CREC_LAZY_WINDOW=${W:-8} python tools/bench_big.py --steps 320 $CREC_BIG_ARGS 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$defs W=${W:-8}:', round(d['us_per_step'],2), 'us/step; window launch', round(d['bpr_triples_lazy']['avg_us'],2), 'us')"
done
