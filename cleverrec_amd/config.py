"""Configuration: the reference's two .properties files merged into one flat dict of strings
(main.py:19-26: ConfigParser lower-cases keys, strips values, '#' starts a full-line comment).
Consumers re-parse the strings exactly like the reference does (model/Recommender.py:16-28);
keys this core adds (adam_mode, sampler_mode, seed) default so that old files keep working."""
import configparser as cp
import os

DEFAULTS = {
    "adam_mode": "tf1",        # tf1 | lazy | deferred  (include/crec.h CREC_ADAM_*)
    "sampler_mode": "mt19937",  # mt19937: the reference's streams, drawn on the device | mt19937_host: by the host form | device: same rule, own generator
}


def read_configs(project_file="./CleverRec.properties", conf_dir=None, recommender=None):
    conf = cp.ConfigParser()
    if not conf.read(project_file, encoding="utf-8"):
        raise FileNotFoundError(project_file)
    configs = dict(conf.items("default"))
    recommender = recommender or configs["recommender"]
    configs["recommender"] = recommender
    conf_dir = conf_dir or configs.get("config_dir", "./conf")
    model_file = os.path.join(conf_dir, recommender + ".properties")
    if not conf.read(model_file, encoding="utf-8"):
        raise FileNotFoundError(model_file)
    configs.update(dict(conf.items("parameters")))
    for k, v in DEFAULTS.items():
        configs.setdefault(k, v)
    return configs


def parse_list(s, typ=float):
    """'[0.7,0.2,0.1]' -> [0.7, 0.2, 0.1]   (the reference's s[1:-1].split(','))"""
    return [typ(x) for x in s.strip()[1:-1].split(",") if x.strip() != ""]


def is_true(s):
    """The reference compares strings to 'True' (RankingRecommender.py:35)."""
    return str(s).strip() == "True"


GRID_KEYS = (("embed_size", int), ("reg", float), ("neg_ratio", int))


def grid(configs):
    """The reference's grid search (main_tuning.py:38-47): embed_size, reg and neg_ratio are bracketed lists in the
    model's .properties file and every combination is run, embed_size outermost, neg_ratio innermost.  Yields the
    {'embed_size': .., 'reg': .., 'neg_ratio': ..} update of each run (values typed as the reference types them).
    A scalar value ('64') counts as a one-element list, so the shipped single-run files work unchanged."""
    axes = []
    for key, typ in GRID_KEYS:
        s = str(configs[key]).strip()
        axes.append(parse_list(s, typ) if s.startswith("[") else [typ(s)])
    for e in axes[0]:
        for r in axes[1]:
            for n in axes[2]:
                yield {"embed_size": e, "reg": r, "neg_ratio": n}


# Shapes this build's kernels are written for (include/crec.h).  The reference (TensorFlow) has no such limits; a
# .properties file beyond them is refused HERE, when the model object is built, with the key, the value and the limit
# -- not as a CREC_EINVAL from the middle of an epoch.
LIMITS = {
    "batch_size": 65536,  # CML, SBPR / CUNE_BPR, LRML, TransCF: the per-step item index comes in at most 64 chunks of 2048 slots
    "batch_size_fm": 65536,  # FM: batch_size * features per row <= 2^20 slots (the libfm files here have up to 16 features per row)
    "batch_size_rows": 1 << 18,  # GMF / MLP / NeuMF, FM: a batch's slots are indexed 16 384 at a time, a larger batch in several chunks
    "batch_size_bpr": 1 << 19,  # BPR: beyond 8192 the batch's index is built in global memory (csrc/index_big.hip); the limit is
                                # what the index descriptors' fields hold
    "topk": 128,          # the fused evaluation ranks 32 candidates per user and pass, up to four passes
    "embed_size": 256,    # table rows are walked as float4 slices by at most 64 lanes
}


def validate(configs, extra_width=0):
    """Raise ValueError for a configuration outside the built shapes.  extra_width: columns a model adds to its rows
    before evaluation (CML: 4)."""
    def bad(key, value, why):
        raise ValueError("%s = %s: %s (this build's limit; see cleverrec_amd/config.py LIMITS and include/crec.h)" % (key, value, why))
    b = int(configs["batch_size"])
    name = configs.get("recommender")
    b_max = LIMITS["batch_size_bpr" if name == "BPR" else ("batch_size_rows" if name in ("GMF", "MLP", "NeuMF", "FM") else "batch_size")]
    if name == "FM":
        b_max = LIMITS["batch_size_fm"]
    if not 1 <= b <= b_max:
        bad("batch_size", b, "batches of 1..%d instances" % b_max)
    ks = parse_list(configs["topk"], int)
    if not ks or ks != sorted(ks) or ks[0] < 1 or ks[-1] > LIMITS["topk"]:
        bad("topk", configs["topk"], "an ascending list of cut-offs in 1..%d" % LIMITS["topk"])
    if "embed_size" in configs and not str(configs["embed_size"]).strip().startswith("["):
        d = int(configs["embed_size"])
        # BPR pads its rows to whole float4 slices itself (zero columns stay zero there); the other models' graphs couple the
        # columns (CML's covariance term, the towers' first layer), so their width has to be a multiple of 4 as given
        pads = configs.get("recommender") == "BPR"
        if d < 1 or (d % 4 and not pads) or (d + 3) // 4 * 4 + extra_width > LIMITS["embed_size"]:
            bad("embed_size", d, ("1..%d" if pads else "a multiple of 4 in 4..%d") % (LIMITS["embed_size"] - extra_width))
