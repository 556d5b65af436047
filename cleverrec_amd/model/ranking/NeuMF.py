"""Neural Collaborative Filtering (reference model/ranking/NeuMF.py): same class name, constructor, hooks;
training is the fused NCF step of csrc/ncf.hip (kind NeuMF)."""
import os

import torch

from ...config import parse_list
from ...runtime import NcfState
from ..RankingRecommender import RankingRecommender


class NeuMF(RankingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(NeuMF, self).__init__(sess, data, configs, logger)
        # get_loss(self.loss_func, self.y, self.logits) (utils/tools.py:66-76): the two names that can train this graph in the
        # reference; 'bpr' ignores the logits (no gradient reaches the model), 'hinge' adds margin = None (TypeError)
        if self.loss_func not in ('cross_entropy', 'square'):
            raise NotImplementedError("NeuMF: loss_func=%s cannot train this graph (utils/tools.py:66-76)" % self.loss_func)
        self.embed_size = int(configs['embed_size'])
        self.layers = parse_list(configs['layers'], int)
        # the class reads reg1/reg2, the shipped conf defines reg_gmf/reg_mlp: accept both
        self.reg1 = float(configs['reg1'] if 'reg1' in configs else configs['reg_gmf'])
        self.reg2 = float(configs['reg2'] if 'reg2' in configs else configs['reg_mlp'])
        if 'gmf_pretrain' in configs and 'mlp_pretrain' in configs:
            # the reference restores before global_variables_initializer (overwritten) and nothing ever saves a
            # checkpoint: the keys are inert there and here
            logger.info(' gmf_pretrain / mlp_pretrain are ignored (no effect in the reference either)')
        logger.info(' model_params: embed_size=%s, layers=%s, reg1=%s, reg2=%s' % (self.embed_size, self.layers, self.reg1, self.reg2)
                    + ', ' + self.model_params)

    def _create_inputs(self):
        self.u_idx = self.i_idx = self.y = None

    def _create_params(self, init=None):
        init = init or {}
        U, I, d, L = self.data.user_nums, self.data.item_nums, self.embed_size, self.layers

        def get(name, shape):
            return init[name] if name in init else self.initializer(shape)
        tables = {'P_gmf': get('P_gmf', [U, d]), 'Q_gmf': get('Q_gmf', [I, d]),
                  'P_mlp': get('P_mlp', [U, L[0] // 2]), 'Q_mlp': get('Q_mlp', [I, L[0] // 2])}
        Ws = [get('W_%d' % k, [L[k], L[k] // 2]) for k in range(len(L))]
        bs = [get('b_%d' % k, [L[k] // 2]) for k in range(len(L))]
        h = get('h_neumf', [d + L[-1] // 2])  # NeuMF.py:51
        self.state = NcfState(self.sess, 'NeuMF', tables, Ws, bs, h, self.lr, reg_gmf=self.reg1, reg_mlp=self.reg2, loss_func=self.loss_func,
                              optimizer=self.state_optimizer)
        self.P_gmf, self.Q_gmf = self.state.t['P_gmf'], self.state.t['Q_gmf']
        self.P_mlp, self.Q_mlp, self.h_neumf = self.state.t['P_mlp'], self.state.t['Q_mlp'], self.state.h
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
        var = {'NeuMF_params/' + k: self.state.t[k].cpu() for k in ('P_gmf', 'Q_gmf', 'P_mlp', 'Q_mlp')}
        var['NeuMF_params/h_neumf'] = self.h_neumf.cpu()
        for k in range(len(self.layers)):
            var['NeuMF_params/W_%d' % k], var['NeuMF_params/b_%d' % k] = self.state.W[k].cpu(), self.state.b[k].cpu()
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
        # pre_scores = sigmoid(logits) (NeuMF.py _predict, rs branch) is the ranking of the logits
        from ... import runtime
        return runtime.ncf_scores(self.sess, self.state, users)
