"""Translational Collaborative Filtering (reference model/ranking/TransCF.py): same class name, constructor and hooks; the
graph is the fused TransCF step of csrc/transcf.hip, the rs / loo scores are its squared distances (ascending: cml_like)."""
import os

import torch

from ...runtime import TranscfState
from ...utils.tools import get_sp_mat
from ..RankingRecommender import RankingRecommender


class TransCF(RankingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(TransCF, self).__init__(sess, data, configs, logger)
        if self.loss_func != 'hinge':  # get_loss(self.loss_func, ui_dist - uj_dist, margin=self.margin) (TransCF.py:57)
            raise NotImplementedError("TransCF: loss_func=%s: the step implements the hinge of the shipped conf (utils/tools.py:72-73)" % self.loss_func)
        if self.is_pairwise != 'True':
            raise NotImplementedError("TransCF trains on (u, i, j) triples: is_pairwise=True (conf/TransCF.properties)")
        self.embed_size, self.reg1, self.reg2, self.margin = int(configs['embed_size']), float(configs['reg1']), float(configs['reg2']), \
            float(configs['margin'])
        logger.info(' model_params: embed_size=%d, reg1=%s, reg2=%s, margin=%s' % (self.embed_size, self.reg1, self.reg2, self.margin)
                    + ', ' + self.model_params)
        # Generate ui_sp_mat and iu_sp_mat (TransCF.py:16), as CSRs
        self.ui_sp_mat, self.iu_sp_mat = None, None
        self._sp_mat = get_sp_mat(data)

    def _create_inputs(self):
        self.u_idx = self.i_idx = self.j_idx = None

    def _create_params(self, init=None):
        init = init or {}
        U, I, d = self.data.user_nums, self.data.item_nums, self.embed_size
        P = init['P'] if 'P' in init else self.initializer([U, d])
        Q = init['Q'] if 'Q' in init else self.initializer([I, d])
        self.state = TranscfState(self.sess, P, Q, self._sp_mat, self.lr, self.reg1, self.reg2, self.margin, optimizer=self.optimizer)
        self.P, self.Q = self.state.P, self.state.Q

    def _create_embeddings(self):
        pass

    def _create_inference(self):
        pass  # neighbourhood aggregation, relation vectors, distances, loss and minimize (TransCF.py:39-62): tcf_* kernels

    def _get_regularizations(self):
        raise NotImplementedError("the regularisers (TransCF.py:65-73) are fused into tcf_triples")

    def _unit_clipping(self):
        pass  # rebinds u_embed / i_embed / j_embed after the train op exists (TransCF.py:62,79-83): no variable is clipped;
        # its one observable effect -- the rs branch of _predict uses clip_1(p_u) -- lives in _eval_scores

    def _predict(self):
        pass

    def _save_model(self):
        os.makedirs(os.path.join(self.saved_model_dir, self.model), exist_ok=True)

    def save(self, path=None):
        """Variables under the reference's checkpoint names (TransCF.py:94)."""
        path = path or os.path.join(self.saved_model_dir, self.model, self.model + '.pt')
        os.makedirs(os.path.dirname(path), exist_ok=True)  # build_model does not call _save_model (TransCF.py:107)
        torch.save({'transcf_params/P': self.P.cpu(), 'transcf_params/Q': self.Q.cpu()}, path)
        return path

    def build_model(self, init=None):
        self._create_inputs()
        self._create_params(init)
        self._create_embeddings()
        self._create_inference()
        self._predict()

    def _train_epoch_device(self, u, i, j):
        return self.state.train_epoch(u, i, j, self.batch_size)

    def _eval_scores(self, users):
        # loo / sampled negatives: pre_scores = ui_dist, built from the unclipped rows (TransCF.py:86-87); rs: the re-bound,
        # row-clipped u_embed against every item (TransCF.py:88-91).  Sorted ascending by the trainer (cml_like)
        loo = self.configs['data.split_way'] == 'loo' or self.neg_samples > 0
        return self.state.scores(users, clip_users=not loo)
