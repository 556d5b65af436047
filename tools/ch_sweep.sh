for ch in 16 32 64; do CREC_CH=$ch python tools/bench_big.py --steps 300 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('ch $ch:', round(d['triples_per_s']/1e6,1), 'Mtr/s', round(d['us_per_step'],2), 'us/step triples', round(d['bpr_triples_lazy']['avg_us'],2), round(d['bpr_triples_lazy']['frac_of_8TBs'],3))"; done
