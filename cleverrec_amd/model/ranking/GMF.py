"""Generalized Matrix Factorization (reference model/ranking/GMF.py): same class name, constructor and
hooks; the graph is the fused NCF step of csrc/ncf.hip (kind GMF)."""
import os

import torch

from ...runtime import NcfState
from ..RankingRecommender import RankingRecommender


class GMF(RankingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(GMF, self).__init__(sess, data, configs, logger)
        self.embed_size = int(configs['embed_size'])
        # the class reads configs['reg'] but the shipped conf defines reg_gmf: accept both
        self.reg = float(configs['reg'] if 'reg' in configs else configs['reg_gmf'])
        # get_loss(self.loss_func, self.y, self.logits) (utils/tools.py:66-76): the two names that can train this graph in the
        # reference; 'bpr' ignores the logits (no gradient reaches the model), 'hinge' adds margin = None (TypeError)
        if self.loss_func not in ('cross_entropy', 'square'):
            raise NotImplementedError("GMF: loss_func=%s cannot train this graph (utils/tools.py:66-76)" % self.loss_func)
        logger.info(' model_params: embed_size=%d, reg=%s' % (self.embed_size, self.reg) + ', ' + self.model_params)

    def _create_inputs(self):
        self.u_idx = self.i_idx = self.y = None  # per-epoch int32 / fp32 device columns

    def _create_params(self, init=None):
        init = init or {}
        U, I, d = self.data.user_nums, self.data.item_nums, self.embed_size
        P = init.get('P', None) if 'P' in init else self.initializer([U, d])
        Q = init.get('Q', None) if 'Q' in init else self.initializer([I, d])
        h = init.get('h_gmf', None) if 'h_gmf' in init else self.initializer([d])  # biases use the same initializer (GMF.py:25-26)
        self.state = NcfState(self.sess, 'GMF', {'P_gmf': P, 'Q_gmf': Q}, [], [], h, self.lr, reg_gmf=self.reg, loss_func=self.loss_func,
                              optimizer=self.state_optimizer)
        self.P, self.Q, self.h_gmf = self.state.t['P_gmf'], self.state.t['Q_gmf'], self.state.h

    def _create_embeddings(self):
        pass

    def _create_inference(self):
        pass

    def _predict(self):
        pass  # pre_scores = sigmoid((pu * Q) . h) (GMF.py:40,57): see _eval_tables

    def _save_model(self):
        os.makedirs(os.path.join(self.saved_model_dir, self.model), exist_ok=True)

    def save(self, path=None):
        path = path or os.path.join(self.saved_model_dir, self.model, self.model + '.pt')
        torch.save({'GMF_params/P': self.P.cpu(), 'GMF_params/Q': self.Q.cpu(), 'GMF_params/h_gmf': self.h_gmf.cpu()}, path)
        return path

    def build_model(self, init=None):
        self._create_inputs()
        self._create_params(init)
        self._create_embeddings()
        self._create_inference()
        self._predict()
        self._save_model()

    def _train_epoch_device(self, u, i, y):
        return self.state.train_epoch(u, i, y, self.batch_size)

    def _eval_tables(self):
        # sigmoid is monotone: ranking by sigmoid((pu*h).q) == ranking by (pu*h).q, which is the GEMM form
        return (self.P * self.h_gmf).contiguous(), self.Q
