"""Trainer base for ranking models (reference model/RankingRecommender.py): same methods and
return shapes -- train_model() -> mean of per-batch summed losses (:33-61), test_model_rs() ->
(HR, MRR, NDCG) dicts of per-user lists keyed by topk index (:198-247), run_model() with the
reference's log lines and best-NDCG@topk[0] bookkeeping (:395-440) -- with the per-batch
sess.run, the numpy argsort and the Python filter/metric loops replaced by libcrec.so calls."""
import math
import time
from collections import defaultdict

import numpy as np
import torch

from .. import runtime
from ..utils.sampler import pairwise_ranking_sampler, pointwise_ranking_sampler
from ..utils.tools import timer
from .Recommender import Recommender


class RankingRecommender(Recommender):
    def __init__(self, sess, data, configs, logger):
        super(RankingRecommender, self).__init__(sess, data, configs, logger)
        self.neg_ratio = int(configs['neg_ratio'])
        self.model_params += ', neg_ratio=%d' % self.neg_ratio
        self.test_users = [int(u) for u in (data.test_users if hasattr(data, 'test_users') else data.ui_test.keys())]
        self.test_batches = math.ceil(len(self.test_users) / self.batch_size_t)
        self._eval_cache = None
        # `sampler_mode` (extension; default mt19937):
        #   mt19937       the reference's index streams bit for bit -- NumPy's MT19937 words, the rejection loops, the shuffle --
        #                 drawn ON the device (runtime.MtSampler, csrc/mt_sampler.hip) with the columns left in HBM; shapes outside
        #                 the kernels' limits (item_nums > 262144) fall to the same stream drawn by the host C++ form
        #   mt19937_host  the host C++ form (utils/sampler.py), same streams
        #   device        same acceptance rule on a counter-based generator: statistically equivalent, NOT stream-compatible
        # `sampler_prefetch` (default True): the NEXT epoch's sample is drawn ahead (device form: enqueued in front of this epoch's
        # steps, the host never waits for it; host form: on a host thread while the GPU trains); the streams and np.random's state
        # afterwards are unchanged
        self.sampler_prefetch = str(configs.get('sampler_prefetch', 'True')).lower() not in ('false', '0', 'no')
        self._dev_sampler = self._mt_sampler = None
        mode = str(configs.get('sampler_mode', 'mt19937')).lower()
        if mode == 'device':
            self._dev_sampler = runtime.DeviceSampler(sess, data, seed=int(configs.get('seed', 0)))
        elif mode == 'mt19937':
            from .._lib import CrecError, EUNSUPPORTED
            try:
                self._mt_sampler = runtime.MtSampler(self.sess, data)
            except CrecError as e:
                if e.code != EUNSUPPORTED:
                    raise
        elif mode != 'mt19937_host':
            raise ValueError("sampler_mode=%s: mt19937 | mt19937_host | device" % mode)

    # -- hooks a model class fills in ---------------------------------------------------------
    def _train_epoch_device(self, u, i, j):
        """Run all batches of one sampled epoch; return per-batch losses (device tensor)."""
        raise NotImplementedError

    def _eval_tables(self):
        """(P_like, Q_like) such that score(u, item) = P_like[u] . Q_like[item] (fused GEMM + top-K path)."""
        raise NotImplementedError

    def _eval_scores(self, users):
        """Models whose score is not a dot product (tower models): a [len(users), I] score matrix, or None."""
        return None

    # -- one epoch (RankingRecommender.py:33-61) -------------------------------------------------
    def train_model(self):
        s = self.sess
        pairwise = self.is_pairwise == 'True'
        if self._dev_sampler is not None:
            ds = self._dev_sampler
            cols = ds.pairwise(self.neg_ratio) if pairwise else ds.pointwise(self.neg_ratio)
            losses = self._train_epoch_device(*cols)
            return float(losses.double().sum().item()) / int(losses.numel())
        if self._mt_sampler is not None:  # the reference's sample, drawn on the device: nothing crosses PCIe
            if self.fism_like:
                raise NotImplementedError("fism_like models are outside this core (SURVEY.md 2.1 #10)")
            cols = self._mt_sampler.next_epoch("pairwise" if pairwise else "pointwise", self.neg_ratio, ahead=self._ahead())
            self._epoch_instances = int(cols[0].numel())
            losses = self._train_epoch_device(*cols)
            return float(losses.double().sum().item()) / math.ceil(int(cols[0].numel()) / self.batch_size)
        if pairwise:
            train_ = pairwise_ranking_sampler(self.data, self.neg_ratio, self.batch_size, fism_like=self.fism_like,
                                              prefetch_next=self._ahead())
            u, i, j = (s.to_dev(a) for a in train_[1:4])
            losses = self._train_epoch_device(u, i, j)
        else:  # pointwise: (u, i, label) instances (RankingRecommender.py:48-58)
            train_ = pointwise_ranking_sampler(self.data, self.neg_ratio, self.batch_size, fism_like=self.fism_like,
                                               prefetch_next=self._ahead())
            u, i = s.to_dev(train_[1]), s.to_dev(train_[2])
            y = s.to_dev(np.asarray(train_[3], dtype=np.float32))
            losses = self._train_epoch_device(u, i, y)
        self._epoch_instances = len(train_[1])
        total_loss = float(losses.double().sum().item())  # total_loss += loss_val over the batches
        return total_loss / train_[0]

    def _ahead(self):
        """draw the next epoch's sample during this one?  Not on run_model's last epoch: nobody would use it"""
        return self.sampler_prefetch and not getattr(self, '_last_epoch', False)

    def _samplers_done(self):
        """end of training: a sample drawn ahead for an epoch that will not come is dropped (np.random is where the last USED sample
        left it, by construction) and nothing keeps running behind the caller's back"""
        if self._mt_sampler is not None:
            self._mt_sampler.drop()
        from ..utils import sampler as host_sampler
        host_sampler.drop_prefetch(join=True)

    # -- full-catalogue evaluation (RankingRecommender.py:198-247) -------------------------------
    def _eval_inputs(self):
        if self._eval_cache is None:
            s, d = self.sess, self.data
            if not hasattr(d, 'seen_csr'):
                from ..data import RankingData
                d = RankingData(d.user_nums, d.item_nums, d.ui_train, d.ui_test)
            self._eval_cache = dict(
                users=s.to_dev(np.asarray(self.test_users, dtype=np.int32)),
                seen_ptr=s.to_dev(d.seen_csr[0]), seen_items=s.to_dev(d.seen_csr[1] if len(d.seen_csr[1]) else np.zeros(1, np.int32)),
                test_ptr=s.to_dev(d.test_csr[0]), test_items=s.to_dev(d.test_csr[1]))
        return self._eval_cache

    def test_model_rs(self):
        HR, MRR, NDCG = defaultdict(list), defaultdict(list), defaultdict(list)
        e = self._eval_inputs()
        Kmax = self.topk[-1]
        tables = None
        for t_id in range(self.test_batches):
            cur = e['users'][t_id * self.batch_size_t:(t_id + 1) * self.batch_size_t]
            sc = self._eval_scores(cur)
            if sc is not None:  # materialised pre_scores (MLP / NeuMF), then argsort + seen filter
                ids, _ = runtime.topk_scores(self.sess, sc, cur, e['seen_ptr'], e['seen_items'], Kmax, self.cml_like)
            else:
                tables = tables or self._eval_tables()
                ids, _ = runtime.eval_topk(self.sess, tables[0], tables[1], cur, e['seen_ptr'], e['seen_items'], Kmax, self.cml_like)
            hr, mrr, ndcg = runtime.rank_metrics(self.sess, ids, cur, self.topk, e['test_ptr'], e['test_items'])
            hr, mrr, ndcg = hr.cpu().numpy(), mrr.cpu().numpy(), ndcg.cpu().numpy()
            for kid in range(len(self.topk)):
                HR[kid].extend(hr[kid].tolist())
                MRR[kid].extend(mrr[kid].tolist())
                NDCG[kid].extend(ndcg[kid].tolist())
        return HR, MRR, NDCG

    def test_model_loo(self):
        """Leave-one-out / sampled-negative evaluation (RankingRecommender.py:250-299): only the items listed in
        ui_test[u] (neg_samples negatives, then the held-out items) compete; the held-out ones are the truth."""
        HR, MRR, NDCG = defaultdict(list), defaultdict(list), defaultdict(list)
        e = self._eval_inputs()
        if 'cand_ptr' not in e:
            d = self.data
            if not hasattr(d, 'loo_csr'):
                from ..data import RankingData
                d = RankingData(d.user_nums, d.item_nums, d.ui_train, d.ui_test)
            (cp, ci), (rp, ri) = d.loo_csr(self.neg_samples)
            s = self.sess
            e.update(cand_ptr=s.to_dev(cp), cand_items=s.to_dev(ci), real_ptr=s.to_dev(rp), real_items=s.to_dev(ri))
        Kmax = self.topk[-1]
        tables = None
        for t_id in range(self.test_batches):
            cur = e['users'][t_id * self.batch_size_t:(t_id + 1) * self.batch_size_t]
            sc = self._eval_scores(cur)
            if sc is not None:
                ids, _ = runtime.topk_scores(self.sess, sc, cur, e['cand_ptr'], e['cand_items'], Kmax, self.cml_like, candidates=True)
            else:
                tables = tables or self._eval_tables()
                ids, _ = runtime.eval_topk(self.sess, tables[0], tables[1], cur, e['cand_ptr'], e['cand_items'], Kmax,
                                           self.cml_like, candidates=True)
            hr, mrr, ndcg = runtime.rank_metrics(self.sess, ids, cur, self.topk, e['real_ptr'], e['real_items'])
            hr, mrr, ndcg = hr.cpu().numpy(), mrr.cpu().numpy(), ndcg.cpu().numpy()
            for kid in range(len(self.topk)):
                HR[kid].extend(hr[kid].tolist())
                MRR[kid].extend(mrr[kid].tolist())
                NDCG[kid].extend(ndcg[kid].tolist())
        return HR, MRR, NDCG

    # -- driver (RankingRecommender.py:395-440) ---------------------------------------------------
    @timer('run_model')
    def run_model(self):
        self.build_model()
        best_ndcg10, best_epoch = 0, 0
        best_metrics = {}
        for epoch in range(self.epoches):
            t1 = time.time()
            self._last_epoch = epoch == self.epoches - 1
            avg_loss = self.train_model()
            dt = time.time() - t1
            self.logger.info(' epoch %d\n  Training loss: %.4f, time: %s' % (epoch + 1, avg_loss, time.strftime('%H:%M:%S', time.gmtime(dt))))
            n_inst = getattr(self, '_epoch_instances', None)
            if n_inst:  # (extension) sampler-inclusive rate of the epoch just run
                self.logger.info('  %.2f M instances/s (%.1f ms, sampling included)' % (n_inst / max(dt, 1e-9) / 1e6, dt * 1e3))
            t2 = time.time()
            if (epoch + 1) % self.T:
                continue
            if self.configs['data.split_way'] == 'loo' or self.neg_samples > 0:
                HR, MRR, NDCG = self.test_model_loo()
            else:
                HR, MRR, NDCG = self.test_model_rs()
            dt2 = time.time() - t2
            self.logger.info('  Testing time: %s' % time.strftime('%H:%M:%S', time.gmtime(dt2)))
            self.logger.info('  %.1f k users/s (%.1f ms)' % (len(self.test_users) / max(dt2, 1e-9) / 1e3, dt2 * 1e3))
            best_flag = False
            for id in range(len(self.topk)):
                hr, mrr, ndcg = np.mean(HR[id]), np.mean(MRR[id]), np.mean(NDCG[id])
                self.logger.info('  (k=%d) HR=%.4f, MRR=%.4f, NDCG=%.4f' % (self.topk[id], hr, mrr, ndcg))
                if id == 0 and ndcg > best_ndcg10:
                    best_flag = True
                    best_ndcg10 = ndcg
                if best_flag:
                    best_metrics[id] = (hr, mrr, ndcg)
                    best_epoch = epoch + 1
        self._samplers_done()
        self.logger.info('best_epoch: %d' % best_epoch)
        for id in range(len(self.topk)):
            if id in best_metrics:  # the reference raises KeyError when NDCG never rose above 0
                hr, mrr, ndcg = best_metrics[id]
                self.logger.info('  (k=%d) HR=%.4f, MRR=%.4f, NDCG=%.4f' % (self.topk[id], hr, mrr, ndcg))
        return best_epoch, best_metrics
