"""Factorization Machine (reference model/rating/FM.py): same class name, constructor and hooks; the
graph is csrc/fm.hip."""
import numpy as np

from ...runtime import FmState
from ..RatingRecommender import RatingRecommender


class FM(RatingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(FM, self).__init__(sess, data, configs, logger)
        self.embed_size, self.reg = int(configs['embed_size']), float(configs['reg'])
        if self.loss_func != 'square':
            raise NotImplementedError("FM is built with loss_func=square (conf/FM.properties)")
        logger.info(' model_params: embed_size=%d, reg=%s' % (self.embed_size, self.reg) + ', ' + self.model_params)

    def _create_inputs(self):
        self.x_idx = self.x_value = self.y = None

    def _create_params(self, init=None):
        init = init or {}
        n = self.data.feature_nums + 1  # 0 as the invalid id (FM.py:30)
        w0 = init['w0'] if 'w0' in init else np.zeros(1, np.float32)  # FM.py:29
        wi = init['wi'] if 'wi' in init else self.initializer([n])
        vif = init['vif'] if 'vif' in init else self.initializer([n, self.embed_size])
        self.state = FmState(self.sess, w0, wi, vif, self.lr, self.reg, optimizer=self.optimizer)
        self.w0, self.wi, self.vif = self.state.w0, self.state.wi, self.state.vif

    def _create_embeddings(self):
        pass

    def _create_inference(self):
        pass

    def _save_model(self):
        pass

    def build_model(self, init=None):
        self._create_inputs()
        self._create_params(init)
        self._create_embeddings()
        self._create_inference()
