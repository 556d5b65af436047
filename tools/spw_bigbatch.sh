#!/bin/bash
# S-big at large batches: the lazy step's 1024-thread geometry (32 slots per workgroup) against the 512-thread one (GPU box)
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj; out=/tmp/crec_spw16big; mkdir -p $out
hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DCREC_LAZY_BIG_B_MAX=1073741824 -c cleverrec_amd/csrc/bpr_train.hip -o $out/bpr_train.o || exit 1
hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $(ls $O/*.o | grep -v "/bpr_train.o") $out/bpr_train.o -L/opt/rocm/lib -lrccl || exit 1
for bb in 32768 65536; do for lib in "" $out/libcrec.so; do
  CREC_LIB=$lib python tools/bench_big.py --batch $bb --steps $(( (1 << 23) / bb )) 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('B=$bb', 'default (16 per workgroup beyond 8192)' if not '$lib' else '32 per workgroup', round(d['us_per_step'],2), 'us/step; kernel', round(d['bpr_triples_lazy']['avg_us'],2))"
done; done
