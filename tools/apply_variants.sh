#!/bin/bash
# Build libcrec.so variants of rows_adam.hip with extra -D flags and time the d = 128 models' steps with each (GPU box).
# usage: tools/apply_variants.sh "-DX=1" "-DY=2" ...
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
n=0
for flags in "$@"; do
  n=$((n+1)); out=/tmp/crec_apply_$n; mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude $flags -c cleverrec_amd/csrc/rows_adam.hip -o $out/rows_adam.o || exit 1
  objs=$(ls $O/*.o | grep -v "/rows_adam.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/rows_adam.o -L/opt/rocm/lib -lrccl || exit 1
  echo "== $flags"; CREC_LIB=$out/libcrec.so python tools/bench_models.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:round(v['us_per_step'],1) for k,v in d.items() if 'us_per_step' in v})"
done
