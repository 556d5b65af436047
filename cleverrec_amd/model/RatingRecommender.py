"""Trainer base for rating models (reference model/RatingRecommender.py:12-105): train_fm() ->
(rmse, mae, avg_loss) with the pre-update predictions, test_fm() -> (rmse, mae), run_model() with the
reference's log lines and best-RMSE bookkeeping."""
import math
import time

import numpy as np
import torch

from .. import runtime
from ..utils.tools import timer
from .Recommender import Recommender


class RatingRecommender(Recommender):
    def __init__(self, sess, data, configs, logger):
        super(RatingRecommender, self).__init__(sess, data, configs, logger)
        self.model = configs['recommender']
        if self.model == 'FM':
            self.is_real_valued = configs['is_real_valued'] == 'True'
        self._dev = None

    def train_model(self):
        pass

    def test_model(self):
        pass

    def _device_data(self):
        if self._dev is None:
            s, d = self.sess, self.data
            self._dev = dict(X_idx_t=s.to_dev(d.X_idx_t), X_value_t=s.to_dev(d.X_value_t), y_t=s.to_dev(d.y_t))
        return self._dev

    def train_fm(self):
        """RatingRecommender.py:26-55: np.random.permutation shuffle, contiguous batches, train RMSE/MAE of the
        pre-update predictions."""
        s, d = self.sess, self.data
        s_idx = np.random.permutation(d.y_tr.size)
        x_idx, x_val, y = s.to_dev(d.X_idx_tr[s_idx]), s.to_dev(d.X_value_tr[s_idx]), s.to_dev(d.y_tr[s_idx])
        y_pre, losses = self.state.train_epoch(x_idx, x_val, y, self.batch_size)
        train_batches = math.ceil(d.y_tr.size / self.batch_size)
        # the reference calls cal_rmse_mae(y_pre, y_tr): symmetric in its arguments
        rmse, mae = runtime.rmse_mae(s, y, y_pre)
        return rmse, mae, float(losses.double().sum().item()) / train_batches

    def test_fm(self):
        dev = self._device_data()
        y_pre = self.state.predict(dev['X_idx_t'], dev['X_value_t'])
        return runtime.rmse_mae(self.sess, dev['y_t'], y_pre)

    @timer('run_model')
    def run_model(self):
        self.build_model()
        best_rmse, best_epoch, best_metrics = float('inf'), 0, []
        for epoch in range(self.epoches):
            t1 = time.time()
            rmse_tr, mae_tr, avg_loss = self.train_fm()
            self.logger.info(' Training epoch %d\n time=%s, RMSE=%.4f, MAE=%.4f' % (epoch + 1, time.strftime('%H:%M:%S', time.gmtime(time.time() - t1)), rmse_tr, mae_tr))
            t2 = time.time()
            rmse_t, mae_t = self.test_fm()
            self.logger.info('  Testing time=%s, RMSE=%.4f, MAE=%.4f' % (time.strftime('%H:%M:%S', time.gmtime(time.time() - t2)), rmse_t, mae_t))
            if rmse_t < best_rmse:
                best_rmse, best_metrics, best_epoch = rmse_t, [rmse_t, mae_t], epoch + 1
        self.logger.info('best_epoch=%d, best_rmse=%.4f, best_mae=%.4f' % (best_epoch, best_metrics[0], best_metrics[1]))
        return best_epoch, best_metrics
