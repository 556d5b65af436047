"""Latent Relational Metric Learning (reference model/ranking/LRML.py): same class name, constructor and hooks; the graph
is the fused LRML step of csrc/lrml.hip, the rs / loo scores are its squared distances (ascending: cml_like)."""
import os

import torch

from ...runtime import LrmlState
from ..RankingRecommender import RankingRecommender


class LRML(RankingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(LRML, self).__init__(sess, data, configs, logger)
        # get_loss(self.loss_func, ui_dist - uj_dist, margin=self.margin) (LRML.py:63): 'hinge' is the one name that uses the
        # margin; 'bpr' would also run in the reference (margin ignored) but is not what the model is
        if self.loss_func != 'hinge':
            raise NotImplementedError("LRML: loss_func=%s: the step implements the hinge of the shipped conf (utils/tools.py:72-73)" % self.loss_func)
        if self.is_pairwise != 'True':
            raise NotImplementedError("LRML trains on (u, i, j) triples: is_pairwise=True (conf/LRML.properties)")
        self.embed_size, self.reg, self.margin = int(configs['embed_size']), float(configs['reg']), float(configs['margin'])
        self.mem_size = int(configs['mem_size'])  # number of memory slots
        logger.info(' model_params: embed_size=%d, mem_size=%d, reg=%s, margin=%s' % (self.embed_size, self.mem_size, self.reg, self.margin)
                    + ', ' + self.model_params)

    def _create_inputs(self):
        self.u_idx = self.i_idx = self.j_idx = None

    def _create_params(self, init=None):
        init = init or {}
        U, I, d, mem = self.data.user_nums, self.data.item_nums, self.embed_size, self.mem_size
        P = init['P'] if 'P' in init else self.initializer([U, d])
        Q = init['Q'] if 'Q' in init else self.initializer([I, d])
        K = init['K'] if 'K' in init else self.initializer([d, mem])
        M = init['M'] if 'M' in init else self.initializer([mem, d])
        self.state = LrmlState(self.sess, P, Q, K, M, self.lr, self.reg, self.margin, optimizer=self.state_optimizer)
        self.P, self.Q, self.K, self.M = self.state.P, self.state.Q, self.state.K, self.state.M

    def _create_embeddings(self):
        pass  # the three gathers (LRML.py:36-40) happen inside lrml_tile

    def _lram(self, user_embed, item_embed, is_test=False):
        raise NotImplementedError("the LRAM module (LRML.py:43-52) is fused into lrml_tile / lrml_scores")

    def _create_inference(self):
        pass

    def _unit_clipping(self):
        pass  # defined by the reference (LRML.py:65-68) but never called by build_model: no variable is ever clipped

    def _predict(self):
        pass

    def _save_model(self):
        os.makedirs(os.path.join(self.saved_model_dir, self.model), exist_ok=True)

    def save(self, path=None):
        """Variables under the reference's checkpoint names (LRML.py:79)."""
        path = path or os.path.join(self.saved_model_dir, self.model, self.model + '.pt')
        os.makedirs(os.path.dirname(path), exist_ok=True)  # build_model does not call _save_model (LRML.py:90)
        torch.save({'lrml_params/P': self.P.cpu(), 'lrml_params/Q': self.Q.cpu(), 'lrml_params/K': self.K.cpu(),
                    'lrml_params/M': self.M.cpu()}, path)
        return path

    def build_model(self, init=None):
        self._create_inputs()
        self._create_params(init)
        self._create_embeddings()
        self._create_inference()
        self._predict()
        # self._save_model()  (commented out in the reference too, LRML.py:90)

    def _train_epoch_device(self, u, i, j):
        return self.state.train_epoch(u, i, j, self.batch_size)

    def _eval_scores(self, users):
        # rs branch: |p_u + r(u, item) - q_item|^2 for every item (LRML.py:74-76); loo / sampled negatives: ui_dist, the same
        # distance at the candidate items (LRML.py:71-72).  Sorted ascending by the trainer (cml_like, RankingRecommender.py:222)
        return self.state.scores(users)
