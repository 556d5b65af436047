# coding: utf-8
"""Driver with the reference's shape (main.py:17-54): read the two .properties files, build the data
object, open the device session, import model.<model_type>.<recommender> by name, run_model()."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cleverrec_amd.config import is_true, read_configs  # noqa: E402
from cleverrec_amd.data import SHAPES, synthetic  # noqa: E402
from cleverrec_amd.utils.tools import get_logger  # noqa: E402

if __name__ == '__main__':
    configs = read_configs('./CleverRec.properties', './conf/')
    recommender = configs['recommender']
    logger = get_logger(configs['log.dir'], recommender)
    logger.info('=' * 100)
    logger.info('Current model: %s' % recommender)

    if configs['model_type'] == 'ranking':
        if configs['data.dataset'] in SHAPES:  # seeded synthetic data with MovieLens marginals
            data = synthetic(configs['data.dataset'])
            logger.info(' Data: dataset=%s, user_nums=%d, item_nums=%d, train=%d' % (configs['data.dataset'], data.user_nums,
                                                                                        data.item_nums, data.n_train))
        else:
            from cleverrec_amd.model.RankingPreprocess import RankingPreprocess
            data = RankingPreprocess(configs, logger)
    else:
        from cleverrec_amd.model.RatingPreprocess import RatingPreprocess
        data = RatingPreprocess(configs, logger)
    if not is_true(configs['gpu.is_gpu']):
        raise RuntimeError('gpu.is_gpu=False: this core has no CPU path (the reference treats the string as truthy anyway)')

    from cleverrec_amd.runtime import Session
    sess = Session(int(configs['gpu.id']))
    module = 'cleverrec_amd.model.' + configs['model_type'] + '.' + recommender
    if importlib.util.find_spec(module) is None:
        raise Exception('Module %s not found.' % module)
    model = getattr(importlib.import_module(module), recommender)(sess, data, configs, logger)
    model.run_model()
