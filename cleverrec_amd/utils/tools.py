"""Host helpers with the reference's names (utils/tools.py:9-87): re_index, timer, get_logger,
get_initializer.  get_loss / get_optimizer have no host object here: loss_func and optimizer
are parsed by the model class and carried out inside the HIP kernels."""
import functools
import logging
import math
import os
import sys
import time

import torch


def re_index(data_set):
    """utils/tools.py:9-15: ids in iteration order -> 0..n-1."""
    return {d: n for n, d in enumerate(data_set)}


def timer(text):
    """utils/tools.py:18-28 (same console lines)."""
    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            t1 = time.time()
            print("Start %s..." % text)
            res = func(*args, **kwargs)
            print("%s done, time: %s" % (text, time.strftime("%H:%M:%S", time.gmtime(time.time() - t1))))
            return res
        return wrapper
    return decorator


def get_logger(log_dir, model):
    """utils/tools.py:31-48: root logger -> <log_dir>/<model>.log + stdout, same format."""
    os.makedirs(log_dir, exist_ok=True)
    logger = logging.getLogger()
    logger.setLevel(logging.DEBUG)
    fmt = logging.Formatter("%(asctime)s  %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    for h in (logging.FileHandler(os.path.join(log_dir, model + ".log")), logging.StreamHandler(sys.stdout)):
        h.setLevel(logging.DEBUG)
        h.setFormatter(fmt)
        logger.addHandler(h)
    return logger


def get_initializer(init_method, stddev=None, generator=None):
    """utils/tools.py:51-63.  Returns f(shape) -> fp32 CPU tensor.  'xavier_uniform' (used by the
    shipped GMF/MLP/NeuMF confs but unhandled by the reference) is Glorot-uniform.
    TF's own Philox stream is not reproducible (and the reference never seeds it); parity tests
    inject tables instead."""
    init_method = init_method.strip()

    def fans(shape):
        if len(shape) < 1:
            return 1, 1
        if len(shape) == 1:
            return shape[0], shape[0]
        return shape[-2], shape[-1]

    def f(shape):
        shape = tuple(int(s) for s in shape)
        g = generator
        if init_method == "normal":
            return torch.empty(shape).normal_(0.0, stddev, generator=g)
        if init_method == "tnormal":  # truncated at 2 sigma, redrawn
            return torch.nn.init.trunc_normal_(torch.empty(shape), 0.0, stddev, -2 * stddev, 2 * stddev, generator=g)
        if init_method == "uniform":
            return torch.empty(shape).uniform_(-stddev, stddev, generator=g)
        fi, fo = fans(shape)
        if init_method in ("xavier", "xavier_uniform"):
            lim = math.sqrt(6.0 / (fi + fo))
            return torch.empty(shape).uniform_(-lim, lim, generator=g)
        if init_method == "xavier_normal":
            # tf.contrib.layers.xavier_initializer(uniform=False) = variance_scaling_initializer(factor=1, FAN_AVG):
            # truncated_normal(0, sqrt(1.3 * factor / n)), n = (fan_in + fan_out) / 2 -- the 1.3 makes up for the truncation
            sd = math.sqrt(1.3 * 2.0 / (fi + fo))
            return torch.nn.init.trunc_normal_(torch.empty(shape), 0.0, sd, -2 * sd, 2 * sd, generator=g)
        raise ValueError("init_method=%r" % init_method)
    return f


def get_SPu(data):
    """utils/tools.py:116-127: per user the items its friends consumed and it did not, as a LIST -- in the order CPython
    iterates the set the reference builds, because ranking_sampler_sbpr indexes that list with the drawn position
    (utils/sampler.py:114-116).  The set arithmetic is therefore the reference's, operation for operation (the
    difference is taken after every friend, which changes the order elements enter the set)."""
    SPu = {}
    for u in data.ui_train:
        u_s_items = set()
        if u in data.user_friends:
            for friend in data.user_friends[u]:
                if friend not in data.ui_train:
                    continue
                u_s_items = u_s_items.union(set(data.ui_train[friend])).difference(set(data.ui_train[u]))
            if u_s_items:
                SPu[u] = list(u_s_items)
    return SPu


def spu_csr(data, SPu):
    """(spu_ptr, spu_items, spu_suk) for crec_sample_sbpr: the SPu lists as rows parallel to ui_train's iteration order,
    and per entry the social coefficient suk = the number of the user's friends (with training items) who consumed it
    (utils/sampler.py:122-129)."""
    import numpy as np
    ptr, items, suk = [0], [], []
    sets = {}
    for u in data.ui_train:
        lst = SPu.get(u, [])
        friends = [f for f in data.user_friends.get(u, []) if f in data.ui_train]
        for f in friends:
            if f not in sets:
                sets[f] = set(data.ui_train[f])
        for it in lst:
            items.append(it)
            suk.append(sum(1 for f in friends if it in sets[f]))  # a friend listed twice counts twice, as in the reference's loop
        ptr.append(len(items))
    return (np.asarray(ptr, np.int64), np.asarray(items if items else [0], np.int32)[:len(items)],
            np.asarray(suk if suk else [0], np.int32)[:len(suk)])


def get_sp_mat(data):
    """utils/tools.py:100-113 (get_sp_mat, TransCF): the row-normalised user-item matrix and its row-normalised transpose as
    two CSRs instead of tf.SparseTensors: (ui_ptr [U+1] int64, ui_items int32, iu_ptr [I+1] int64, iu_users int32).
    Row r of the first holds ui_train[r] in list order (every entry weighs 1/len(ui_train[r])); row i of the second the users
    that hold item i, in ui_train's iteration order (every entry weighs 1/count).  The weights themselves are implied by
    the row lengths (float32(1.0 / n), the SparseTensor's values) and formed on the device."""
    import numpy as np
    U, I = data.user_nums, data.item_nums
    ui_len = np.zeros(U, dtype=np.int64)
    for u, items in data.ui_train.items():
        ui_len[u] = len(items)
    ui_ptr = np.zeros(U + 1, dtype=np.int64)
    np.cumsum(ui_len, out=ui_ptr[1:])
    ui_items = np.empty(int(ui_ptr[-1]), dtype=np.int32)
    us, its = [], []
    for u, items in data.ui_train.items():
        ui_items[ui_ptr[u]:ui_ptr[u + 1]] = np.asarray(items, dtype=np.int32)
        us.append(np.full(len(items), u, dtype=np.int32))
        its.append(np.asarray(items, dtype=np.int32))
    us = np.concatenate(us) if us else np.zeros(0, np.int32)
    its = np.concatenate(its) if its else np.zeros(0, np.int32)
    order = np.argsort(its, kind="stable")  # by item, ties in ui_train's iteration order
    iu_users = np.ascontiguousarray(us[order])
    iu_ptr = np.zeros(I + 1, dtype=np.int64)
    np.cumsum(np.bincount(its, minlength=I), out=iu_ptr[1:])
    return ui_ptr, ui_items, iu_ptr, iu_users
