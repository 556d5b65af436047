"""FM training / prediction / RMSE (HIP, through the C ABI) against the torch-autograd restatement of
model/rating/FM.py with TF1 dense Adam (oracle/models_torch.py).  Needs an MI355X."""
import numpy as np
import pytest

from oracle import models_torch as MT
from cleverrec_amd import runtime
from cleverrec_amd.utils.metrics import cal_rmse_mae
from _adam_close import assert_adam_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


@pytest.mark.parametrize("d,F", [(32, 2), (8, 3), (64, 5), (16, 13), (32, 40)])  # beyond 8 features per row: the form that walks them twice
def test_fm_steps_match_oracle(sess, d, F):
    g = np.random.Generator(np.random.PCG64(d + F))
    n_feat, B, N = 300, 256, 3 * 256 + 100
    w0, wi, vif = np.array([0.1], np.float32), (g.standard_normal(n_feat) * 0.1).astype(np.float32), \
        (g.standard_normal((n_feat, d)) * 0.1).astype(np.float32)
    x_idx = g.integers(0, n_feat, (N, F)).astype(np.int32)
    x_val = np.where(g.random((N, F)) < 0.7, 1.0, g.random((N, F)) * 2).astype(np.float32)
    y = g.integers(1, 6, N).astype(np.float32)
    st = runtime.FmState(sess, w0, wi, vif, lr=1e-3, reg=1e-3)
    om = MT.FM(w0.reshape(()), wi, vif, lr=1e-3, reg=1e-3)
    y_pre, loss = st.train_epoch(sess.to_dev(x_idx), sess.to_dev(x_val), sess.to_dev(y), B)
    ol, oy = [], []
    for b in range(len(loss)):
        sl = slice(b * B, (b + 1) * B)
        ol.append(om.step(x_idx[sl], x_val[sl], y[sl]))
        oy.append(om.last_y_pre)
    np.testing.assert_allclose(loss.cpu().numpy(), ol, rtol=1e-5)
    np.testing.assert_allclose(y_pre.cpu().numpy(), np.concatenate(oy), rtol=1e-5, atol=1e-6)
    for dev, name in ((st.w0, "w0"), (st.wi, "wi"), (st.vif, "vif")):
        assert_adam_close(om, name, dev, *st.slots[name])
    # prediction + RMSE/MAE against the reference-shaped Python loop
    yp = st.predict(sess.to_dev(x_idx), sess.to_dev(x_val))
    rm, ma = runtime.rmse_mae(sess, sess.to_dev(y), yp)
    r2, m2 = cal_rmse_mae(y.astype(np.float64), yp.cpu().numpy().astype(np.float64))
    assert rm == pytest.approx(r2, rel=1e-12) and ma == pytest.approx(m2, rel=1e-12)


def test_fm_batch_beyond_one_index_chunk(sess):
    """batch_size * features beyond the 16 384 slots a workgroup sorts in LDS (the reference takes any batch_size): the batch's index
    comes in several chunks which the update launch walks in order.  C4's fields, B = 20 000 (40 000 slots: three chunks), two
    full batches and a ragged one."""
    g = np.random.Generator(np.random.PCG64(45))
    U, I, d, B = 3000, 2000, 32, 20000
    n_feat, N = U + I, 2 * B + 777
    w0, wi, vif = np.array([0.0], np.float32), (g.standard_normal(n_feat + 1) * 0.01).astype(np.float32), \
        (g.standard_normal((n_feat + 1, d)) * 0.01).astype(np.float32)
    x_idx = np.stack([g.integers(0, U, N), U + np.minimum(g.zipf(1.2, N) - 1, I - 1)], 1).astype(np.int32)
    x_val = np.ones((N, 2), np.float32)
    y = g.integers(1, 6, N).astype(np.float32)
    st = runtime.FmState(sess, w0, wi, vif, lr=1e-3, reg=1e-3)
    om = MT.FM(w0.reshape(()), wi, vif, lr=1e-3, reg=1e-3)
    y_pre, loss = st.train_epoch(sess.to_dev(x_idx), sess.to_dev(x_val), sess.to_dev(y), B)
    ol = [om.step(x_idx[b * B:(b + 1) * B], x_val[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss.cpu().numpy(), ol, rtol=1e-5)
    for dev, name in ((st.w0, "w0"), (st.wi, "wi"), (st.vif, "vif")):
        assert_adam_close(om, name, dev, *st.slots[name])


def test_fm_at_the_c4_shape(sess):
    """BASELINE configs[3] shape: two one-hot fields (user, U + item), N = 9 746 features, d = 32, B = 6 144 -- one
    epoch of four batches (the last one ragged), every variable under the rounding envelope, losses within 1e-5."""
    g = np.random.Generator(np.random.PCG64(44))
    U, I, d, B = 6040, 3706, 32, 6144
    n_feat, N = U + I, 3 * B + 1000
    w0, wi, vif = np.array([0.0], np.float32), (g.standard_normal(n_feat + 1) * 0.01).astype(np.float32), \
        (g.standard_normal((n_feat + 1, d)) * 0.01).astype(np.float32)
    users = g.integers(0, U, N)
    items = np.minimum(g.zipf(1.2, N) - 1, I - 1)
    x_idx = np.stack([users, U + items], 1).astype(np.int32)
    x_val = np.ones((N, 2), np.float32)
    y = g.integers(1, 6, N).astype(np.float32)
    st = runtime.FmState(sess, w0, wi, vif, lr=1e-3, reg=1e-3)
    om = MT.FM(w0.reshape(()), wi, vif, lr=1e-3, reg=1e-3)
    y_pre, loss = st.train_epoch(sess.to_dev(x_idx), sess.to_dev(x_val), sess.to_dev(y), B)
    ol, oy = [], []
    for b in range(len(loss)):
        sl = slice(b * B, (b + 1) * B)
        ol.append(om.step(x_idx[sl], x_val[sl], y[sl]))
        oy.append(om.last_y_pre)
    np.testing.assert_allclose(loss.cpu().numpy(), ol, rtol=1e-5)
    np.testing.assert_allclose(y_pre.cpu().numpy(), np.concatenate(oy), rtol=1e-5, atol=1e-6)
    for dev, name in ((st.w0, "w0"), (st.wi, "wi"), (st.vif, "vif")):
        assert_adam_close(om, name, dev, *st.slots[name])


def test_fm_class_surface(sess, tmp_path):
    """FM(sess, data, configs, logger) on a small libfm-ish dataset written to disk: RatingPreprocess reader,
    train_fm()/test_fm() against the oracle on the same np.random.permutation stream."""
    import logging
    import torch
    from cleverrec_amd.model.RatingPreprocess import RatingPreprocess
    from cleverrec_amd.model.rating.FM import FM
    g = np.random.Generator(np.random.PCG64(1))
    U, I = 40, 25
    root = tmp_path / "toy"
    root.mkdir()
    for name, n in (("toy.train.libfm", 900), ("toy.test.libfm", 200)):
        with open(root / name, "w") as f:
            for _ in range(n):
                f.write("%d,%d:1,%d:1\n" % (g.integers(1, 6), g.integers(1, U + 1), U + g.integers(1, I + 1)))
    configs = {"recommender": "FM", "data.root_dir": str(tmp_path), "data.dataset": "toy", "train": ".train.libfm",
               "test": ".test.libfm", "is_real_valued": "True", "epoches": "2", "batch_size": "256",
               "test.batch_size": "128", "lr": "0.01", "test.neg_samples": "0", "is_pairwise": "False",
               "init_method": "normal", "stddev": "0.01", "loss_func": "square", "optimizer": "Adam",
               "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]", "embed_size": "32", "reg": "0.001"}
    data = RatingPreprocess(configs, logging.getLogger("t"))
    assert data.X_idx_tr.shape == (900, 2) and data.feature_nums == len(set(data.X_idx_tr.ravel()) | set(data.X_idx_t.ravel()))
    m = FM(sess, data, configs, logging.getLogger("t"))
    n = data.feature_nums + 1
    init = {"w0": np.zeros(1, np.float32), "wi": (g.standard_normal(n) * 0.01).astype(np.float32),
            "vif": (g.standard_normal((n, 32)) * 0.01).astype(np.float32)}
    m.build_model(init)
    om = MT.FM(np.float32(0.0), init["wi"], init["vif"], lr=0.01, reg=0.001)
    np.random.seed(5)
    rmse, mae, avg_loss = m.train_fm()
    np.random.seed(5)
    s_idx = np.random.permutation(900)
    xi, xv, yy = data.X_idx_tr[s_idx], data.X_value_tr[s_idx], data.y_tr[s_idx]
    tot, preds = 0.0, []
    for b in range(4):
        sl = slice(b * 256, (b + 1) * 256)
        tot += om.step(xi[sl], xv[sl], yy[sl])
        preds.append(om.last_y_pre)
    r2, m2 = cal_rmse_mae(np.concatenate(preds).astype(np.float64), yy.astype(np.float64))
    assert avg_loss == pytest.approx(tot / 4, rel=1e-5) and rmse == pytest.approx(r2, rel=1e-5) and mae == pytest.approx(m2, rel=1e-5)
    rt, mt = m.test_fm()
    w = {k: v for k, v in om.params.items()}
    yp = om.y_pre(w, data.X_idx_t, data.X_value_t).numpy()
    r3, m3 = cal_rmse_mae(data.y_t.astype(np.float64), yp.astype(np.float64))
    assert rt == pytest.approx(r3, rel=1e-4) and mt == pytest.approx(m3, rel=1e-4)


@pytest.mark.parametrize("optimizer", ["SGD", "Adagrad"])
def test_fm_sgd_and_adagrad_match_oracle(sess, optimizer):
    """configs['optimizer'] = SGD | Adagrad on the FM graph (utils/tools.py:79-87): all three variables are dense (the l2
    term reads the full tables), so every row moves every step."""
    from oracle import models_torch as MT
    g = np.random.Generator(np.random.PCG64(17))
    nf, d, F, B = 120, 16, 3, 128
    N = 2 * B + 30
    w0, wi, vif = np.array([0.1], np.float32), (g.standard_normal(nf) * 0.1).astype(np.float32), (g.standard_normal((nf, d)) * 0.1).astype(np.float32)
    x_idx = g.integers(0, nf, (N, F)).astype(np.int32)
    x_val = np.where(g.random((N, F)) < 0.7, 1.0, g.random((N, F)) * 2).astype(np.float32)
    y = g.integers(1, 6, N).astype(np.float32)
    lr = 0.01
    st = runtime.FmState(sess, w0, wi, vif, lr=lr, reg=1e-3, optimizer=optimizer)
    om = MT.FM(w0.reshape(()), wi, vif, lr=lr, reg=1e-3, optimizer=optimizer)
    y_pre, loss = st.train_epoch(sess.to_dev(x_idx), sess.to_dev(x_val), sess.to_dev(y), B)
    # the loss is a sum of (y - y_hat)^2 over the batch plus the full-table l2 terms: an fp32 sum in ANY order is only an
    # approximation of it, so both sides are held against the float64 value at the oracle's weights -- within 1e-5
    ol64, ol = [], []
    for b in range(len(loss)):
        sl = slice(b * B, (b + 1) * B)
        ol64.append(om.loss64(x_idx[sl], x_val[sl], y[sl]))
        ol.append(om.step(x_idx[sl], x_val[sl], y[sl]))
    np.testing.assert_allclose(loss.cpu().numpy(), ol64, rtol=1e-5)
    np.testing.assert_allclose(ol, ol64, rtol=1e-5)
    for dev, name in ((st.w0, "w0"), (st.wi, "wi"), (st.vif, "vif")):
        ref = om.numpy(name)
        tol = om.bounds(name)[0] + 1e-6 * np.abs(ref) + 1e-7
        assert (np.abs(dev.cpu().numpy().reshape(ref.shape) - ref) <= tol).all(), name
    assert np.abs(st.wi.cpu().numpy() - wi).max() > 1e-3
