"""Social BPR (reference model/ranking/SBPR.py): same class name, constructor and hooks; train_model is rebound to
train_model_sbpr like the reference (SBPR.py:18).  `data` must carry `user_friends` (RankingPreprocess with social_file)."""
import os

import torch

from ...runtime import SbprState
from ...utils.sampler import ranking_sampler_sbpr
from ...utils.tools import get_SPu
from ..RankingRecommender import RankingRecommender


class SBPR(RankingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(SBPR, self).__init__(sess, data, configs, logger)
        if self.loss_func != 'bpr':  # SBPR.py:52 calls get_loss(self.loss_func, x) with one argument: only 'bpr' can run
            raise NotImplementedError("SBPR: loss_func=%s cannot train this graph (utils/tools.py:66-76)" % self.loss_func)
        self.embed_size, self.reg = int(configs['embed_size']), float(configs['reg'])
        logger.info(' model_params: embed_size=%d, reg=%s' % (self.embed_size, self.reg) + ', ' + self.model_params)
        self.SPu = get_SPu(data)  # SBPR.py:16
        self.train_model = self.train_model_sbpr  # SBPR.py:18

    def _create_inputs(self):
        self.u_idx = self.i_idx = self.i_s_idx = self.i_neg_idx = self.suk = None

    def _create_params(self, init=None):
        init = init or {}
        U, I, d = self.data.user_nums, self.data.item_nums, self.embed_size
        P = init['P'] if 'P' in init else self.initializer([U, d])
        Q = init['Q'] if 'Q' in init else self.initializer([I, d])
        bias = init['bias'] if 'bias' in init else torch.zeros(I + 1)  # tf.zeros(item_nums + 1), SBPR.py:37
        self.state = SbprState(self.sess, P, Q, bias, self.lr, self.reg, optimizer=self.state_optimizer)
        self.P, self.Q, self.bias = self.state.P, self.state.Q, self.state.bias

    def _create_embeddings(self):
        pass  # the four gathers and three biased scores (SBPR.py:39-50) happen inside sbpr_instances

    def _create_inference(self):
        pass  # loss and optimizer.minimize (SBPR.py:52-55)

    def _predict(self):
        pass

    def _save_model(self):
        os.makedirs(os.path.join(self.saved_model_dir, self.model), exist_ok=True)

    def save(self, path=None):
        """Variables under the reference's checkpoint names (SBPR.py:63-64)."""
        path = path or os.path.join(self.saved_model_dir, self.model, self.model + '.pt')
        torch.save({'sbpr_params/P': self.P.cpu(), 'sbpr_params/Q': self.Q.cpu(), 'sbpr_params/bias': self.bias.cpu()}, path)
        return path

    def build_model(self, init=None):
        self._create_inputs()
        self._create_params(init)
        self._create_embeddings()
        self._create_inference()
        self._predict()
        self._save_model()

    def train_model_sbpr(self, is_suk=True):
        """RankingRecommender.train_model_sbpr (model/RankingRecommender.py:101-116)."""
        train_ = ranking_sampler_sbpr(self.data, self.SPu, self.neg_ratio, self.batch_size, is_suk=True, prefetch_next=self._ahead())
        s = self.sess
        cols = [s.to_dev(train_[c]) for c in range(1, 6)]
        losses = self.state.train_epoch(*cols, self.batch_size)
        return float(losses.double().sum().item()) / train_[0]

    def _eval_tables(self):
        # rs branch: pre_scores = u_embed @ Q^T, no bias (SBPR.py:60-61).  loo / sampled negatives: ui_scores = p.q_i + b_i
        # (SBPR.py:58-59): the same dot product over rows widened by the bias column
        if self.configs['data.split_way'] == 'loo' or self.neg_samples > 0:
            U, I = self.P.shape[0], self.Q.shape[0]
            pad = torch.zeros((U, 3), device=self.P.device)
            Pa = torch.cat([self.P, torch.ones((U, 1), device=self.P.device), pad], 1).contiguous()
            Qa = torch.cat([self.Q, self.bias[:I, None], torch.zeros((I, 3), device=self.Q.device)], 1).contiguous()
            return Pa, Qa
        return self.state.P, self.state.Q
