# coding: utf-8
"""Grid search with the reference's shape (main_tuning.py:17-65): the two .properties files, one data object, then one
model per (embed_size, reg, neg_ratio) combination of the bracketed lists in the model's file, each built and run with
run_model().  The device session is opened once; every model owns its tables."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cleverrec_amd.config import grid, is_true, read_configs  # noqa: E402
from cleverrec_amd.data import SHAPES, synthetic  # noqa: E402
from cleverrec_amd.utils.tools import get_logger  # noqa: E402


def run_grid(sess, data, configs, logger):
    """-> [(update, run_model()'s return value)] in grid order."""
    recommender = configs['recommender']
    module = 'cleverrec_amd.model.' + configs['model_type'] + '.' + recommender
    if importlib.util.find_spec(module) is None:
        raise Exception('Module %s not found.' % module)
    cls = getattr(importlib.import_module(module), recommender)
    results = []
    for update in grid(configs):
        run_configs = dict(configs)
        run_configs.update(update)  # numbers, as in the reference; the model classes re-parse them
        logger.info('Grid point: %s' % update)
        results.append((update, cls(sess, data, run_configs, logger).run_model()))
    return results


if __name__ == '__main__':
    configs = read_configs('./CleverRec.properties', './conf/')
    logger = get_logger(configs['log.dir'], configs['recommender'])
    logger.info('=' * 100)
    logger.info('Current model: %s' % configs['recommender'])
    if configs['model_type'] == 'ranking':
        if configs['data.dataset'] in SHAPES:
            data = synthetic(configs['data.dataset'])
        else:
            from cleverrec_amd.model.RankingPreprocess import RankingPreprocess
            data = RankingPreprocess(configs, logger)
    else:
        from cleverrec_amd.model.RatingPreprocess import RatingPreprocess
        data = RatingPreprocess(configs, logger)
    if not is_true(configs['gpu.is_gpu']):
        raise RuntimeError('gpu.is_gpu=False: this core has no CPU path')
    from cleverrec_amd.runtime import Session
    run_grid(Session(int(configs['gpu.id'])), data, configs, logger)
