"""Base Recommender (reference model/Recommender.py:9-40): same constructor, same parsed
attributes; `sess` is a cleverrec_amd.runtime.Session (or None: one is opened on gpu.id)."""
from ..config import parse_list, validate
from ..runtime import Session
from ..utils.tools import get_initializer


class Recommender(object):
    def __init__(self, sess, data, configs, logger):
        self.model = configs['recommender']
        if sess is None:
            sess = Session(int(configs.get('gpu.id', 0)))
        self.sess, self.data, self.configs, self.logger = sess, data, configs, logger
        self._get_common_params()

    def _get_common_params(self):
        c = self.configs
        # shapes outside the built kernels are refused here.  Evaluation widens the rows by 4 columns for CML (distances as dot
        # products) and for SBPR / CUNE_BPR's loo / sampled-negatives branch (the bias column): embed_size must leave room
        name = c.get('recommender')
        loo = c.get('data.split_way') == 'loo' or int(c.get('test.neg_samples', 0) or 0) > 0
        validate(c, extra_width=4 if name == 'CML' or (name in ('SBPR', 'CUNE_BPR') and loo) else 0)
        self.epoches, self.batch_size, self.batch_size_t = int(c['epoches']), int(c['batch_size']), int(c['test.batch_size'])
        self.lr, self.neg_samples = float(c['lr']), int(c['test.neg_samples'])
        # presence flags, as in the reference (Recommender.py:19)
        self.fism_like, self.cml_like = 'fism_like' in c, 'cml_like' in c
        self.is_pairwise = c['is_pairwise']  # the *string* 'True'/'False' (RankingRecommender.py:35)
        self.initializer = get_initializer(c['init_method'], float(c['stddev']))
        # the reference attaches l2_regularizer(0.001) to every variable but never adds
        # REGULARIZATION_LOSSES to the loss, so it has no effect: nothing to carry.
        self.loss_func = c['loss_func']
        # get_optimizer (utils/tools.py:79-87): 'SGD' | 'Adam' | 'Adagrad'; any other name leaves the reference with
        # optim = None and an AttributeError at .minimize -- here a ValueError up front
        self.optimizer = c['optimizer']
        if self.optimizer not in ('SGD', 'Adam', 'Adagrad'):
            raise ValueError("optimizer=%s: utils/tools.py get_optimizer knows SGD, Adam, Adagrad" % self.optimizer)
        # 'tf1' (default) | 'lazy' for Adam; SGD / Adagrad update touched rows only, which is their TF1 semantics
        self.adam_mode = {'Adam': c.get('adam_mode', 'tf1'), 'SGD': 'sgd', 'Adagrad': 'adagrad'}[self.optimizer]
        if name == 'BPR' and self.adam_mode == 'tf1' and self.batch_size > 8192:
            # the swept TF1 step indexes a batch in LDS (at most 8192 triples); beyond that the same results come from the deferred
            # form: every row replays the steps it sat out when a batch touches it, indexed in global memory
            self.adam_mode = 'deferred'
            self.logger.info(' batch_size=%d > 8192: Adam runs in its deferred form (TF1 results, touched rows only)' % self.batch_size)
        # GMF / MLP / NeuMF, LRML and SBPR / CUNE_BPR take adam_mode=lazy as well: their states call it the optimizer 'LazyAdam'
        self.state_optimizer = 'LazyAdam' if self.optimizer == 'Adam' and self.adam_mode == 'lazy' else self.optimizer
        self.saved_model_dir = c['saved_dir']
        self.T = int(c['test.interval'])
        self.topk = parse_list(c['topk'], int)
        self.model_params = 'lr=%s, loss_func=%s' % (self.lr, c['loss_func'])

    def build_model(self):
        raise NotImplementedError

    def train_model(self):
        raise NotImplementedError

    def test_model(self):
        raise NotImplementedError

    def run_model(self):
        raise NotImplementedError
