"""Rating data for FM (reference model/RatingPreprocess.py): same class name and attributes.

The FM input contract is the comma-separated libfm-ish text the reference's utils/fm_to_libfm.py writes:
    label,idx:val,idx:val,...
Feature strings are mapped to dense ids in first-seen order, train file first, then test
(RatingPreprocess.py:56-85); feature_nums = number of distinct feature strings."""
import os

import numpy as np

from ..utils.tools import timer


class RatingPreprocess(object):
    def __init__(self, configs, logger):
        self.configs, self.logger = configs, logger
        self.model = configs['recommender']
        self.file_path = os.path.join(configs['data.root_dir'], configs['data.dataset'])
        if self.model != 'FM':
            raise NotImplementedError('the rating path serves FM (the only rating model the reference implements)')
        self.is_real_valued = configs['is_real_valued'] == 'True'
        if not self.is_real_valued:
            raise NotImplementedError('is_real_valued=False: FM.py references an undefined x_value in that branch')
        self.X_idx_tr, self.X_value_tr, self.y_tr, self.X_idx_t, self.X_value_t, self.y_t = self._load_data()
        self.feature_nums = len(self.all_features)
        del self.all_features

    @timer('Load data')
    def _load_data(self):
        self.all_features = {}
        train_file = os.path.join(self.file_path, self.configs['data.dataset'] + self.configs['train'])
        test_file = os.path.join(self.file_path, self.configs['data.dataset'] + self.configs['test'])
        X_idx_tr, X_value_tr, y_tr = self._read_file(train_file)
        X_idx_t, X_value_t, y_t = self._read_file(test_file)
        return X_idx_tr, X_value_tr, y_tr, X_idx_t, X_value_t, y_t

    def _read_file(self, file_name):
        """-> (int32 [n,F] dense feature ids, float32 [n,F] values, float32 [n] labels); rows must share F
        (the reference turns the lists into rectangular numpy arrays, RatingRecommender.py:28)."""
        feats = self.all_features
        idx, val, lab = [], [], []
        with open(file_name, 'r') as fr:
            for line in fr:
                lst = line.strip().split(',')
                if lst == ['']:
                    continue
                lab.append(float(lst[0]))
                ri, rv = [], []
                for col in lst[1:]:
                    f, v = col.split(':')
                    if f not in feats:
                        feats[f] = len(feats)
                    ri.append(feats[f])
                    rv.append(float(v))
                idx.append(ri)
                val.append(rv)
        F = {len(r) for r in idx}
        if len(F) > 1:
            raise ValueError('%s: rows have different feature counts %s' % (file_name, sorted(F)))
        return np.array(idx, dtype=np.int32), np.array(val, dtype=np.float32), np.array(lab, dtype=np.float32)
