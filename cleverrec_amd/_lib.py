"""ctypes binding of libcrec.so (include/crec.h).  No CPU fallback: if the library or a
symbol is missing this module raises, and device entry points fail with the library's own
error text when no gfx950 is present."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CREC_LIB") or os.path.join(_HERE, "lib", "libcrec.so")  # CREC_LIB: an alternative build

OK, EINVAL, EHIP, ENOMEM, EUNSUPPORTED, ENODEVICE = 0, -1, -2, -3, -4, -5
ADAM_TF1, ADAM_LAZY, OPT_SGD, OPT_ADAGRAD, ADAM_DEFERRED = 0, 1, 2, 3, 4
_CODES = {EINVAL: "CREC_EINVAL", EHIP: "CREC_EHIP", ENOMEM: "CREC_ENOMEM", EUNSUPPORTED: "CREC_EUNSUPPORTED",
          ENODEVICE: "CREC_ENODEVICE"}


class CrecError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s: %s" % (_CODES.get(code, code), text))
        self.code = code


class BprTables(C.Structure):
    _fields_ = [("P", C.c_void_p), ("Q", C.c_void_p), ("mP", C.c_void_p), ("vP", C.c_void_p), ("mQ", C.c_void_p),
                ("vQ", C.c_void_p), ("U", C.c_int64), ("I", C.c_int64), ("d", C.c_int32)]


class Adam(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("reg", C.c_float),
                ("adam_mode", C.c_int32), ("beta1_power", C.c_float), ("beta2_power", C.c_float), ("step", C.c_int64)]


class ShardStats(C.Structure):
    _fields_ = [("bytes_ids_sent", C.c_int64), ("bytes_rows_sent", C.c_int64), ("bytes_grads_sent", C.c_int64),
                ("rows_requested", C.c_int64), ("rows_served", C.c_int64), ("max_rows_served_per_batch", C.c_int64),
                ("steps", C.c_int64)]


class BprDeferred(C.Structure):
    _fields_ = [("lastP", C.c_void_p), ("lastQ", C.c_void_p)]


class Param(C.Structure):
    _fields_ = [("var", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("rows", C.c_int64), ("cols", C.c_int32)]


NCF_GMF, NCF_MLP, NCF_NEUMF = 0, 1, 2
LOSS_CROSS_ENTROPY, LOSS_SQUARE = 0, 1


class Ncf(C.Structure):
    _fields_ = [("kind", C.c_int32), ("P_gmf", Param), ("Q_gmf", Param), ("P_mlp", Param), ("Q_mlp", Param),
                ("n_layers", C.c_int32), ("small", Param), ("reg_gmf", C.c_float), ("reg_mlp", C.c_float), ("loss_func", C.c_int32)]


class Fm(C.Structure):
    _fields_ = [("w0", Param), ("wi", Param), ("vif", Param), ("reg", C.c_float)]


class Sbpr(C.Structure):
    _fields_ = [("P", Param), ("Q", Param), ("bias", Param), ("reg", C.c_float)]


class Cune(C.Structure):
    _fields_ = [("P", Param), ("Q", Param), ("bias", Param), ("s", Param), ("reg", C.c_float)]


class Lrml(C.Structure):
    _fields_ = [("P", Param), ("Q", Param), ("small", Param), ("mem_size", C.c_int32), ("reg", C.c_float), ("margin", C.c_float)]


class Transcf(C.Structure):
    _fields_ = [("P", Param), ("Q", Param), ("ui_ptr", C.c_void_p), ("ui_items", C.c_void_p), ("iu_ptr", C.c_void_p),
                ("iu_users", C.c_void_p), ("reg1", C.c_float), ("reg2", C.c_float), ("margin", C.c_float)]


class Cml(C.Structure):
    _fields_ = [("P", Param), ("Q", Param), ("reg", C.c_float), ("margin", C.c_float)]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_SIGNATURES = {
    "crec_version": (C.c_int, []),
    "crec_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "crec_ctx_destroy": (None, [_vp]),
    "crec_last_error": (C.c_char_p, [_vp]),
    "crec_ctx_set_option": (C.c_int, [_vp, C.c_char_p, _i64]),
    "crec_profile_enable": (C.c_int, [_vp, C.c_int]),
    "crec_profile_read": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "crec_profile_read_units": (C.c_int, [_vp, C.c_int, C.POINTER(_i64)]),
    "crec_rng_seed": (C.c_int, [_vp, C.c_uint32]),
    "crec_rng_set_state": (C.c_int, [_vp, _vp, _i32]),
    "crec_rng_get_state": (C.c_int, [_vp, _vp, C.POINTER(_i32)]),
    "crec_rng_randint": (C.c_int, [_vp, _i64, C.POINTER(_i64)]),
    "crec_rng_permutation": (C.c_int, [_vp, _i64, _vp]),
    "crec_sample_pairwise": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, C.c_int, _vp, _vp, _vp, C.POINTER(_i64)]),
    "crec_sample_pointwise": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, C.c_int, _vp, _vp, _vp, C.POINTER(_i64)]),
    "crec_sample_cml": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, C.c_int, _vp, _vp, _vp, C.POINTER(_i64)]),
    "crec_sample_sbpr": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    "crec_sample_negatives_device": (C.c_int, [_vp, C.c_uint64, C.c_uint64, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "crec_dsampler_create": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, C.POINTER(_vp), _vp]),
    "crec_dsampler_destroy": (None, [_vp]),
    "crec_dsampler_sample": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _i32, _vp, _vp, _vp, _vp]),
    "crec_dsampler_finish": (C.c_int, [_vp, _vp, C.POINTER(_i64), _vp, C.POINTER(_i32)]),
    "crec_dsampler_stats": (C.c_int, [_vp, _vp]),
    "crec_debug_mt_jump": (C.c_int, [_vp, C.c_int, _vp]),
    "crec_bpr_train_epoch": (C.c_int, [_vp, C.POINTER(BprTables), C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_bpr_train_epoch_deferred": (C.c_int, [_vp, C.POINTER(BprTables), C.POINTER(BprDeferred), C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_bpr_adam_flush": (C.c_int, [_vp, C.POINTER(BprTables), C.POINTER(BprDeferred), C.POINTER(Adam), _vp]),
    "crec_ncf_train_epoch": (C.c_int, [_vp, C.POINTER(Ncf), C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_ncf_scores": (C.c_int, [_vp, C.POINTER(Ncf), _vp, _i32, _vp, _vp]),
    "crec_topk_scores": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "crec_eval_topk_candidates": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "crec_topk_scores_candidates": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "crec_fm_train_epoch": (C.c_int, [_vp, C.POINTER(Fm), C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "crec_fm_predict": (C.c_int, [_vp, C.POINTER(Fm), _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_rmse_mae": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "crec_cml_train_epoch": (C.c_int, [_vp, C.POINTER(Cml), C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "crec_sbpr_train_epoch": (C.c_int, [_vp, C.POINTER(Sbpr), C.POINTER(Adam), _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_cune_train_epoch": (C.c_int, [_vp, C.POINTER(Cune), C.POINTER(Adam), _vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_lrml_train_epoch": (C.c_int, [_vp, C.POINTER(Lrml), C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_lrml_scores": (C.c_int, [_vp, C.POINTER(Lrml), _vp, _i32, _vp, _vp]),
    "crec_transcf_train_epoch": (C.c_int, [_vp, C.POINTER(Transcf), C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_transcf_scores": (C.c_int, [_vp, C.POINTER(Transcf), _vp, _i32, _i32, _vp, _vp, _vp]),
    "crec_cml_eval_tables": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp]),
    "crec_bpr_train_epoch_dp": (C.c_int, [_vp, C.POINTER(BprTables), C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "crec_comm_unique_id": (C.c_int, [_vp, _vp]),
    "crec_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "crec_comm_destroy": (C.c_int, [_vp]),
    "crec_comm_init_local": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "crec_bpr_train_epoch_sharded": (C.c_int, [_vp, C.POINTER(BprTables), _i64, C.POINTER(Adam), _vp, _vp, _vp, _i64, _i32, _vp,
                                               C.POINTER(ShardStats), _vp]),
    "crec_comm_peer_create": (C.c_int, [_vp, _i64, _vp]),
    "crec_comm_peer_connect": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "crec_comm_peer_connect_local": (C.c_int, [C.POINTER(_vp), C.c_int, _i64]),
    "crec_comm_set_timeout": (C.c_int, [_vp, C.c_double]),
    "crec_comm_peer_fences": (C.c_int, [_vp, C.c_int]),
    "crec_comm_status": (C.c_int, [_vp, _vp]),
    "crec_debug_alltoallv": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "crec_debug_allreduce": (C.c_int, [_vp, _vp, _i64, _vp]),
    "crec_shard_plan_host": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _vp, C.POINTER(_i32), _vp, _vp, _vp, _vp]),
    "crec_shard_owner_index_host": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, C.POINTER(_i32)]),
    "crec_debug_shard_plan": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "crec_debug_lazy_index": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "crec_debug_shard_owner_index": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "crec_eval_topk": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "crec_rank_metrics": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load():
    """dlopen libcrec.so and bind every symbol of include/crec.h (raises if any is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s not built: run `python -m cleverrec_amd.build` (needs hipcc)" % LIB_PATH)
        # torch first: its wheel bundles its own libamdhip64.so.7 / libhsa-runtime64; libcrec.so must
        # bind to that already-loaded runtime (device pointers and streams are shared with torch).
        # Loaded the other way round the process ends up with two HSA runtimes and no visible GPU.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the export is missing
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def _np(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One crec_ctx.  device=None -> host-only (RNG + samplers)."""

    def __init__(self, device=None):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.crec_ctx_create(-1 if device is None else int(device), C.byref(h))
        if rc != OK:
            raise CrecError(rc, "crec_ctx_create(device=%r): %s" % (device, self.lib.crec_last_error(None).decode()))
        self.h, self.device = h, device

    def close(self):
        if getattr(self, "h", None):
            self.lib.crec_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def check(self, rc):
        if rc != OK:
            raise CrecError(rc, self.lib.crec_last_error(self.h).decode())

    def set_option(self, name, value):
        """crec_ctx_set_option: tuning / test knobs ("ch", "wide", "wide_round", "eval_shape", "eval_no_bitmap", "ncf_generic", "no_zero_copy", "shard_unfused", "dsampler_general", "dsampler_budget", "flow", "flow_steps", "flow_deal", "lazy_window")."""
        self.check(self.lib.crec_ctx_set_option(self.h, name.encode(), int(value)))

    # ---- measurement ---------------------------------------------------------------------------
    KERNELS = ("bpr_index_batches", "bpr_step_fused", "bpr_triples_lazy", "bpr_steps_flow", "bpr_reduce_loss", "eval_sweep",
               "eval_merge", "rank_metrics", "unused8", "bpr_adam_dense", "rows_index_batches", "rows_apply", "small_adam_dense", "ncf_tile", "fm_instances", "cml_instances", "ncf_pair_logits",
               "shard_plan_batches", "shard_serve_rows", "shard_owner_apply", "peer_allreduce_adam")

    def profile(self, on):
        self.check(self.lib.crec_profile_enable(self.h, int(bool(on))))

    def profile_read(self):
        """{kernel: (total_ms, launches)} accumulated since profile(True)."""
        out = {}
        for k, name in enumerate(self.KERNELS):
            ms, n = C.c_double(), C.c_int64()
            self.check(self.lib.crec_profile_read(self.h, k, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out

    def profile_units(self, name):
        """work units of the counted launches of `name` (bpr_steps_flow: steps; every other kernel: one per launch)"""
        n = C.c_int64()
        self.check(self.lib.crec_profile_read_units(self.h, self.KERNELS.index(name), C.byref(n)))
        return n.value

    # ---- RNG ----------------------------------------------------------------------------------
    def seed(self, s):
        self.check(self.lib.crec_rng_seed(self.h, int(s) & 0xFFFFFFFF))

    def set_state(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        assert key.shape == (624,)
        self.check(self.lib.crec_rng_set_state(self.h, _np(key), int(pos)))

    def get_state(self):
        key, pos = np.empty(624, dtype=np.uint32), C.c_int32()
        self.check(self.lib.crec_rng_get_state(self.h, _np(key), C.byref(pos)))
        return key, pos.value

    def randint(self, n):
        out = C.c_int64()
        self.check(self.lib.crec_rng_randint(self.h, int(n), C.byref(out)))
        return out.value

    def permutation(self, n):
        out = np.empty(int(n), dtype=np.int64)
        self.check(self.lib.crec_rng_permutation(self.h, int(n), _np(out)))
        return out

    # ---- samplers (host) ------------------------------------------------------------------------
    alloc = staticmethod(np.empty)  # (n, dtype) -> ndarray the samplers write into; utils/sampler.py swaps in page-locked memory

    def _sample(self, fn, csr, item_nums, neg_ratio, cap, third):
        keys, ptr, items = csr
        u, i = self.alloc(max(cap, 1), np.int32), self.alloc(max(cap, 1), np.int32)
        n = C.c_int64()
        self.check(fn(self.h, _np(ptr), _np(items), len(keys), _np(keys), int(item_nums), int(neg_ratio), _np(u), _np(i),
                      _np(third), C.byref(n)))
        return n.value, u, i

    def sample_pairwise(self, csr, item_nums, neg_ratio):
        cap = int(csr[1][-1]) * neg_ratio
        j = self.alloc(max(cap, 1), np.int32)
        n, u, i = self._sample(self.lib.crec_sample_pairwise, csr, item_nums, neg_ratio, cap, j)
        return u[:n], i[:n], j[:n]

    def sample_pointwise(self, csr, item_nums, neg_ratio):
        cap = int(csr[1][-1]) * (1 + neg_ratio)
        y = self.alloc(max(cap, 1), np.float32)
        n, u, i = self._sample(self.lib.crec_sample_pointwise, csr, item_nums, neg_ratio, cap, y)
        return u[:n], i[:n], y[:n]

    def sample_sbpr(self, csr, item_nums, neg_ratio, spu_csr):
        """ranking_sampler_sbpr: spu_csr = (spu_ptr int64 [n_users+1], spu_items int32, spu_suk int32), rows parallel to csr's."""
        keys, ptr, items = csr
        spu_ptr, spu_items, spu_suk = spu_csr
        cap = int(ptr[-1]) * neg_ratio
        u, i, k, j = (self.alloc(max(cap, 1), np.int32) for _ in range(4))
        suk = self.alloc(max(cap, 1), np.float32)
        n = C.c_int64()
        self.check(self.lib.crec_sample_sbpr(self.h, _np(ptr), _np(items), len(keys), _np(keys), int(item_nums), int(neg_ratio),
                                             _np(spu_ptr), _np(spu_items), _np(spu_suk), _np(u), _np(i), _np(k), _np(j), _np(suk),
                                             C.byref(n)))
        n = n.value
        return u[:n], i[:n], k[:n], j[:n], suk[:n]

    def sample_cml(self, csr, item_nums, neg_ratio):
        cap = int(csr[1][-1])
        neg = self.alloc(max(cap * neg_ratio, 1), np.int32)
        n, u, i = self._sample(self.lib.crec_sample_cml, csr, item_nums, neg_ratio, cap, neg)
        return u[:n], i[:n], neg[:n * neg_ratio].reshape(n, neg_ratio)
