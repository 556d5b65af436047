"""Collaborative User Network Embedding + BPR (reference model/ranking/CUNE_BPR.py): same class name, constructor and
hooks; train_model is train_model_sbpr(is_suk=False) like the reference (CUNE_BPR.py:22).  The latent friends come from
utils/tools.get_topK_friends_and_SPu (network, deep walks, word2vec, cosine top-K) unless `data` already carries
`user_friends` and `SPu` (e.g. computed once and reused across a grid search)."""
import os

import torch

from ...runtime import CuneState
from ...utils.sampler import ranking_sampler_sbpr
from ...utils.tools import get_topK_friends_and_SPu
from .SBPR import SBPR


class CUNE_BPR(SBPR):
    def __init__(self, sess, data, configs, logger, friends_and_SPu=None):
        super(SBPR, self).__init__(sess, data, configs, logger)  # RankingRecommender's constructor (SBPR's derives SPu from the trust file's friends)
        if self.loss_func != 'bpr':  # CUNE_BPR.py:58 calls get_loss(self.loss_func, x) with one argument: only 'bpr' can run
            raise NotImplementedError("CUNE_BPR: loss_func=%s cannot train this graph (utils/tools.py:66-76)" % self.loss_func)
        self.embed_size, self.reg = int(configs['embed_size']), float(configs['reg'])
        self.walk_count, self.walk_length, self.walk_dim, self.window_size, self.topK_f = (
            int(configs['walk_count']), int(configs['walk_length']), int(configs['walk_dim']), int(configs['window_size']), int(configs['topk_f']))
        logger.info(' model_params: embed_size=%d, reg=%s, walk_count=%d, walk_length=%d, walk_dim=%d, window_size=%d, topK_f=%d'
                    % (self.embed_size, self.reg, self.walk_count, self.walk_length, self.walk_dim, self.window_size, self.topK_f) + ', ' + self.model_params)
        if friends_and_SPu is None:  # CUNE_BPR.py:20
            friends_and_SPu = get_topK_friends_and_SPu(data, self.walk_count, self.walk_length, self.walk_dim, self.window_size, self.topK_f)
        self.data.user_friends, self.SPu = friends_and_SPu
        self.train_model = self.train_model_cune  # partial(self.train_model_sbpr, is_suk=False), CUNE_BPR.py:22

    def _create_params(self, init=None):
        init = init or {}
        U, I, d = self.data.user_nums, self.data.item_nums, self.embed_size
        P = init['P'] if 'P' in init else self.initializer([U, d])
        Q = init['Q'] if 'Q' in init else self.initializer([I, d])
        bias = init['bias'] if 'bias' in init else torch.zeros(I + 1)  # tf.zeros(item_nums + 1), CUNE_BPR.py:41
        self.state = CuneState(self.sess, P, Q, bias, init.get('s', 0.0), self.lr, self.reg, optimizer=self.state_optimizer)  # s = 0.0, :42
        self.P, self.Q, self.bias, self.s = self.state.P, self.state.Q, self.state.bias, self.state.s

    def save(self, path=None):
        """Variables under the reference's checkpoint names (CUNE_BPR.py:68-69: `s` is not in the Saver's var_list)."""
        path = path or os.path.join(self.saved_model_dir, self.model, self.model + '.pt')
        torch.save({'cune_params/P': self.P.cpu(), 'cune_params/Q': self.Q.cpu(), 'cune_params/bias': self.bias.cpu()}, path)
        return path

    def train_model_cune(self):
        """RankingRecommender.train_model_sbpr(is_suk=False) (model/RankingRecommender.py:101-116)."""
        train_ = ranking_sampler_sbpr(self.data, self.SPu, self.neg_ratio, self.batch_size, is_suk=False, prefetch_next=self._ahead())
        s = self.sess
        losses = self.state.train_epoch(*(s.to_dev(train_[c]) for c in range(1, 5)), self.batch_size)
        return float(losses.double().sum().item()) / train_[0]
