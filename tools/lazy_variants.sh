#!/bin/bash
# S-big: build libcrec.so variants of the lazy triple pass (finisher loads in flight) and sweep the chunk length (GPU box).
# usage: tools/lazy_variants.sh "DEEP CH CH ..." ...   e.g. tools/lazy_variants.sh "8 16 24" "11 16 24 32"
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
for v in "$@"; do
  set -- $v; deep=$1; shift; out=/tmp/crec_lazy_d${deep}_$(echo $CREC_DEFS | tr -c 'A-Za-z0-9' _); mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DCREC_LAZY_DEEP=$deep $CREC_DEFS \
        -c cleverrec_amd/csrc/bpr_train.hip -o $out/bpr_train.o || exit 1
  objs=$(ls $O/*.o | grep -v "/bpr_train.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/bpr_train.o -L/opt/rocm/lib -lrccl || exit 1
  for ch in "$@"; do CREC_LIB=$out/libcrec.so CREC_CH=$ch python tools/bench_big.py --steps 300 $CREC_BIG_ARGS 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('deep $deep $CREC_DEFS ch $ch:', round(d['triples_per_s']/1e6,1), 'Mtr/s', round(d['us_per_step'],2), 'us/step kernel', round(d['bpr_triples_lazy']['avg_us'],2), 'us', round(d['step_algorithmic_bytes']/d['bpr_triples_lazy']['avg_us']/1e3/8000,3), 'of 8 TB/s')"; done
done
