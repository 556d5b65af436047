"""utils/metrics.py of the reference, same names.  The batched device version is
crec_rank_metrics (csrc/eval_topk.hip); these host forms serve single calls and RMSE/MAE."""
import math


def cal_ranking_metrics(real_items, rec_items, K):
    """utils/metrics.py:9-19: (hit/min(K,len(real)), sum of 1/rank over hits, dcg/idcg) with the
    idcg taken over all true items."""
    rec = [int(x) for x in rec_items]
    hit, mrr, dcg, idcg = 0, 0, 0, 0
    for idx_true, item in enumerate(real_items):
        if item in rec:
            hit += 1
            idx = rec.index(item)
            mrr += 1.0 / (idx + 1)
            dcg += 1.0 / math.log2(idx + 2)
        idcg += 1.0 / math.log2(idx_true + 2)
    return hit / min(K, len(real_items)), mrr, dcg / idcg


def cal_rmse_mae(y, y_pre):
    """utils/metrics.py:22-29."""
    abs_sum, square_sum = 0, 0
    for a, b in zip(y, y_pre):
        res = a - b
        abs_sum += abs(res)
        square_sum += res ** 2
    return math.sqrt(square_sum / len(y)), abs_sum / len(y)
