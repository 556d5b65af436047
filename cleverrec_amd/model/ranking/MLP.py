"""Multi-layer Perceptron of NCF (reference model/ranking/MLP.py): same class name, constructor, hooks;
training is the fused NCF step of csrc/ncf.hip (kind MLP)."""
import os

import torch

from ...config import parse_list
from ...runtime import NcfState
from ..RankingRecommender import RankingRecommender


class MLP(RankingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(MLP, self).__init__(sess, data, configs, logger)
        # get_loss(self.loss_func, self.y, self.logits) (utils/tools.py:66-76): the two names that can train this graph in the
        # reference; 'bpr' ignores the logits (no gradient reaches the model), 'hinge' adds margin = None (TypeError)
        if self.loss_func not in ('cross_entropy', 'square'):
            raise NotImplementedError("MLP: loss_func=%s cannot train this graph (utils/tools.py:66-76)" % self.loss_func)
        self.layers = parse_list(configs['layers'], int)
        self.reg = float(configs['reg'] if 'reg' in configs else configs['reg_mlp'])
        logger.info(' model_params: layers=%s, reg=%s' % (self.layers, self.reg) + ', ' + self.model_params)

    def _create_inputs(self):
        self.u_idx = self.i_idx = self.y = None

    def _create_params(self, init=None):
        init = init or {}
        U, I, L = self.data.user_nums, self.data.item_nums, self.layers

        def get(name, shape):
            return init[name] if name in init else self.initializer(shape)
        P, Q = get('P', [U, L[0] // 2]), get('Q', [I, L[0] // 2])
        Ws = [get('W_%d' % k, [L[k], L[k] // 2]) for k in range(len(L))]  # MLP.py:34-37
        bs = [get('b_%d' % k, [L[k] // 2]) for k in range(len(L))]
        h = get('h_mlp', [L[-1] // 2])
        for k in range(1, len(L)):
            if L[k] != L[k - 1] // 2:
                raise ValueError("layers=%s: layer k must take the previous layer's output width" % L)
        self.state = NcfState(self.sess, 'MLP', {'P_mlp': P, 'Q_mlp': Q}, Ws, bs, h, self.lr, reg_mlp=self.reg, loss_func=self.loss_func,
                              optimizer=self.state_optimizer)
        self.P, self.Q, self.h_mlp = self.state.t['P_mlp'], self.state.t['Q_mlp'], self.state.h
        self.mlp_params = {}
        for k in range(len(L)):
            self.mlp_params['W_%d' % k], self.mlp_params['b_%d' % k] = self.state.W[k], self.state.b[k]

    def _create_embeddings(self):
        pass

    def _create_inference(self):
        pass

    def _predict(self):
        pass

    def _save_model(self):
        os.makedirs(os.path.join(self.saved_model_dir, self.model), exist_ok=True)

    def save(self, path=None):
        path = path or os.path.join(self.saved_model_dir, self.model, self.model + '.pt')
        var = {'MLP_params/P': self.P.cpu(), 'MLP_params/Q': self.Q.cpu(), 'MLP_params/h_mlp': self.h_mlp.cpu()}
        for k in range(len(self.layers)):
            var['MLP_params/W_%d' % k], var['MLP_params/b_%d' % k] = self.state.W[k].cpu(), self.state.b[k].cpu()
        torch.save(var, path)
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

    def _eval_scores(self, users):
        # logits of every (test user, item) pair through the tower; sigmoid is monotone, so the ranking of
        # pre_scores = sigmoid(logits) (MLP.py _predict, rs branch) is the ranking of the logits
        from ... import runtime
        return runtime.ncf_scores(self.sess, self.state, users)
