"""Collaborative Metric Learning (reference model/ranking/CML.py): same class name, constructor and hooks;
train_model is rebound to train_model_cml exactly like the reference (CML.py:16)."""
import math

from ...runtime import CmlState
from ...utils.sampler import ranking_sampler_cml
from ..RankingRecommender import RankingRecommender


class CML(RankingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(CML, self).__init__(sess, data, configs, logger)
        self.embed_size, self.reg, self.margin = int(configs['embed_size']), float(configs['reg']), float(configs['margin'])
        logger.info(' model_params: embed_size=%d, reg=%s, margin=%s' % (self.embed_size, self.reg, self.margin) + ', ' + self.model_params)
        self.train_model = self.train_model_cml  # CML.py:16

    def _create_inputs(self):
        self.u_idx = self.i_idx = self.neg_items = None

    def _create_params(self, init=None):
        init = init or {}
        P = init['P'] if 'P' in init else self.initializer([self.data.user_nums, self.embed_size])
        Q = init['Q'] if 'Q' in init else self.initializer([self.data.item_nums, self.embed_size])
        self.state = CmlState(self.sess, P, Q, self.lr, self.reg, self.margin, optimizer=self.optimizer)
        self.P, self.Q = self.state.P, self.state.Q

    def _create_embeddings(self):
        pass

    def _create_inference(self):
        pass  # _unit_clipping only rebinds graph tensors after the train op exists: no variable is ever clipped
        # (SURVEY 2.3); its one observable effect -- rs-eval uses row-clipped user vectors -- lives in eval_tables

    def _predict(self):
        pass

    def build_model(self, init=None):
        self._create_inputs()
        self._create_params(init)
        self._create_embeddings()
        self._create_inference()
        self._predict()

    def train_model_cml(self):
        """RankingRecommender.train_model_cml (model/RankingRecommender.py:90-100)."""
        if self._dev_sampler is not None:
            u, i, neg = self._dev_sampler.cml(self.neg_ratio)
            losses = self.state.train_epoch(u, i, neg, self.batch_size)
            return float(losses.double().sum().item()) / int(losses.numel())
        if self._mt_sampler is not None:  # the reference's sample, drawn on the device
            u, i, neg = self._mt_sampler.next_epoch("cml", self.neg_ratio, ahead=self._ahead())
            self._epoch_instances = int(u.numel())
            losses = self.state.train_epoch(u, i, neg, self.batch_size)
            return float(losses.double().sum().item()) / math.ceil(int(u.numel()) / self.batch_size)
        train_ = ranking_sampler_cml(self.data, self.neg_ratio, self.batch_size, prefetch_next=self._ahead())
        s = self.sess
        losses = self.state.train_epoch(s.to_dev(train_[1]), s.to_dev(train_[2]), s.to_dev(train_[3]), self.batch_size)
        return float(losses.double().sum().item()) / train_[0]

    def _eval_tables(self):
        # with cml_like in the conf the trainer sorts ascending (:222-223).  CML._predict: the loo / sampled-negatives
        # branch scores with u_i_dist, built from the UNCLIPPED p_u before _unit_clipping rebinds u_embed (CML.py:42,81-82);
        # only the rs branch sees clip_1(p_u) (CML.py:76,84)
        loo = self.configs['data.split_way'] == 'loo' or self.neg_samples > 0
        return self.state.eval_tables(clip_users=not loo)
