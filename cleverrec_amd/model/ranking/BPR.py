"""Bayesian Personalized Ranking (reference model/ranking/BPR.py): same class name, constructor
and hook methods; the TF1 graph those hooks built is the fused HIP step of csrc/bpr_train.hip."""
import os

import torch

from ...runtime import BprState
from ..RankingRecommender import RankingRecommender


class BPR(RankingRecommender):
    def __init__(self, sess, data, configs, logger):
        super(BPR, self).__init__(sess, data, configs, logger)
        self.embed_size, self.reg = int(configs['embed_size']), float(configs['reg'])
        if self.loss_func != 'bpr':
            raise NotImplementedError("BPR is built with loss_func=bpr (conf/BPR.properties)")
        logger.info(' model_params: embed_size=%d, reg=%s' % (self.embed_size, self.reg) + ', ' + self.model_params)

    def _create_inputs(self):
        # u_idx / i_idx / j_idx placeholders (BPR.py:16-21) become int32 device columns per epoch
        self.u_idx = self.i_idx = self.j_idx = None

    def _create_params(self, P0=None, Q0=None):
        # 'BPR_params/P' [user_nums, d], 'BPR_params/Q' [item_nums, d] (BPR.py:23-29)
        P0 = self.initializer([self.data.user_nums, self.embed_size]) if P0 is None else P0
        Q0 = self.initializer([self.data.item_nums, self.embed_size]) if Q0 is None else Q0
        self.state = BprState(self.sess, P0, Q0, self.lr, self.reg, self.adam_mode)
        d = self.embed_size  # (the state pads the rows to whole float4 slices; the variables are the reference's [*, embed_size])
        self.P, self.Q = self.state.P[:, :d], self.state.Q[:, :d]

    def _create_embeddings(self):
        pass  # the three embedding_lookups (BPR.py:31-34) happen inside bpr_step_fused / bpr_triples_lazy

    def _create_inference(self):
        pass  # scores, loss and optimizer.minimize (BPR.py:36-44) are the fused step

    def _predict(self):
        pass  # pre_scores = u_embed @ Q^T (BPR.py:51) is crec_eval_topk's GEMM

    def _save_model(self):
        tmp_dir = os.path.join(self.saved_model_dir, self.model)
        os.makedirs(tmp_dir, exist_ok=True)

    def save(self, path=None):
        """Variables under the reference's checkpoint names (BPR.py:53-58)."""
        path = path or os.path.join(self.saved_model_dir, self.model, self.model + '.pt')
        self.state.flush()
        torch.save({'BPR_params/P': self.P.cpu().contiguous(), 'BPR_params/Q': self.Q.cpu().contiguous()}, path)
        return path

    def build_model(self, P0=None, Q0=None):
        self._create_inputs()
        self._create_params(P0, Q0)
        self._create_embeddings()
        self._create_inference()
        self._predict()
        self._save_model()

    def _train_epoch_device(self, u, i, j):
        return self.state.train_epoch(u, i, j, self.batch_size)

    def _eval_tables(self):
        self.state.flush()  # adam_mode deferred: every row replays the steps it sat out; a no-op otherwise
        return self.state.P, self.state.Q
