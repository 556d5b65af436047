#!/bin/bash
# Build libcrec.so variants of bpr_train.hip with different tuning / ablation macros and bench each (GPU box).
# usage: tools/variants.sh "UNR RPW PIPE [ABL [NT [WT [BLOCK [ITEMS_FIRST]]]]]" ...   e.g. tools/variants.sh "2 2 0" "2 2 0 1" ; BIG=1 adds the S-big leg
cd "$(dirname "$0")/.." || exit 1
O=cleverrec_amd/lib/obj
for v in "$@"; do
  set -- $v; tag="u$1_r$2_p$3_a${4:-0}_n${5:-0}_w${6:-0}_b${7:-256}_i${8:-0}"; out=/tmp/crec_$tag; mkdir -p $out
  hipcc --offload-arch=gfx950 -x hip -O3 -fPIC -std=c++17 -ffp-contract=off -Iinclude -DCREC_UNR=$1 -DCREC_RPW=$2 -DCREC_PIPE=$3 -DCREC_ABL=${4:-0} -DCREC_NT=${5:-0} -DCREC_FUSED_WT=${6:-0} -DCREC_FUSED_BLOCK=${7:-256} -DCREC_ITEMS_FIRST=${8:-0} \
        -c cleverrec_amd/csrc/bpr_train.hip -o $out/bpr_train.o -Rpass-analysis=kernel-resource-usage 2> $out/res.txt || { tail -5 $out/res.txt; exit 1; }
  objs=$(ls $O/*.o | grep -v "/bpr_train.o")
  hipcc --offload-arch=gfx950 -shared -o $out/libcrec.so $objs $out/bpr_train.o -L/opt/rocm/lib -lrccl || exit 1
  occ=$(grep -A12 "bpr_step_fusedILi16" $out/res.txt | grep -m1 "Occupancy" | sed 's/.*: //; s/ \[.*//')
  vg=$(grep -A12 "bpr_step_fusedILi16" $out/res.txt | grep -m1 " VGPRs:" | sed 's/.*VGPRs: //; s/ \[.*//')
  CREC_LIB=$out/libcrec.so python bench.py --cpu-steps 0 --no-big 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$tag vgpr $vg occ $occ', round(d['value']/1e6,1), 'Mtr/s', round(d['ms_per_step']*1e3,2), 'us/step', {k:round(v['avg_us'],2) for k,v in d['kernels'].items() if k!='bpr_index_batches'})"
  if [ -n "$BIG" ]; then CREC_LIB=$out/libcrec.so python tools/bench_big.py --steps 200 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('   big $tag', round(d['triples_per_s']/1e6,1), 'Mtr/s', round(d['us_per_step'],1), 'us/step user', round(d['bpr_triples_lazy']['avg_us'],1), 'alg GB/s', round(d['step_algorithmic_GBs']))"; fi
done
