#!/bin/bash
# S-big: sweep the heavy-row chunk length (CREC_CH) of the lazy step (GPU box)
cd "$(dirname "$0")/.." || exit 1
for ch in 8 12 16 24 32 48; do CREC_CH=$ch python tools/bench_big.py --steps 300 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('ch $ch:', round(d['triples_per_s']/1e6,1), 'Mtr/s', round(d['us_per_step'],2), 'us/step user', round(d['bpr_triples_lazy']['avg_us'],2), 'item', round(d['bpr_rows_lazy']['avg_us'],2))"; done
