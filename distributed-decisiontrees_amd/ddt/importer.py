"""Model importer: trained tree ensembles -> the reference's perfect-heap wire format.

The reference's model compiler is not published; its engine only accepts PERFECT binary trees in heap order
with `feature < threshold` going left (rtl/DTEngine/core/DTPU.sv:20-28 capacity note, :594-596 child rule,
:637,659-661 flag bits).  This module produces exactly those two streams (weights lines + feature-index
lines, packing of rtl/DTEngine/core/PipelinedMUX.sv:65) from

  * scikit-learn regressors / classifiers (DecisionTree*, RandomForest*, ExtraTrees*, GradientBoosting*),
  * XGBoost JSON dumps (`Booster.save_model("m.json")`), parsed without the xgboost package.

Rules applied:
  pad_to_perfect   a leaf above depth D becomes a dummy sub-tree whose leaves all repeat its value (SURVEY A10b:
                   the published RTL's early-leaf flag is not usable).
  `<=` -> `<`      scikit-learn goes left iff float32(x) <= t (t is float64).  The emitted fp32 threshold is
                   nextafter(largest fp32 <= t, +inf), so that  x <= t  <=>  x < thr  for every fp32 x.
                   XGBoost already uses `<`.
  missing          default direction -> bit 13 of the feature-index entry ("missing goes right"); the missing
                   pattern is the canonical quiet NaN 0x7FC00000 (the engine tests bit equality, DTPU.sv:653:
                   callers must present NaNs in that canonical form).
  comparator       real models have negative features, where the reference's raw-bit signed-int comparator
                   (DTPU.sv:655) is inverted; imported models therefore set cmp_mode = 1 (IEEE '<').
  leaf values      rounded to fp32 and brought into the engine's EXACT LEAF DOMAIN (include/ddt.h, option leaf_domain_check: +0 or
                   a normal with 2^-102 <= |v| < 2^96 -- on it no partial sum of the reference-order reduction can be sub-normal,
                   overflow or be -0, which the reference's adder treats differently from IEEE-754:
                   FPAdder_2cycles_latency.v:313-320,376-385): -0, sub-normals and every |v| < 2^-102 are flushed to +0 (an
                   absolute error below 2^-102 per leaf); a leaf with |v| >= 2^96, an infinity or a NaN raises ValueError
                   -- such a model does not load with the default options, and silently clamping it would change its scores.
  sparse=True      instead of padding to a perfect heap (2^(D+1) words per tree -- hopeless for a depth-16 random
                   forest), emit the SPARSE stream of include/ddt.h (ddt_load_model_sparse): one 128-bit line per internal
                   node {threshold, feature entry | leaf flags, left, right} in breadth-first order.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np

from .engine import (MISSING_DEFAULT, findex_lines_per_tree, make_params, make_sparse_params,
                     weights_lines_per_tree)

MAX_LEVELS = 16  # CSR205 num_levels is 4 bits (EngineCSR.sv:230)


@dataclass
class ImportedModel:
    """Wire-format model + what the engine needs to score it."""

    wlines: np.ndarray        # uint32
    flines: np.ndarray        # uint16
    num_trees: int
    num_levels: int
    num_features: int
    num_classes: int = 1      # > 1: one-vs-all, tree i belongs to class i % num_classes (interleaved)
    base_score: np.ndarray = field(default_factory=lambda: np.zeros(1, np.float64))  # per class, added by the caller
    missing_bits: int = MISSING_DEFAULT
    cmp_mode: int = 1
    node_lines: np.ndarray | None = None       # sparse=True: uint32 [n_lines, 4]; wlines / flines are then empty
    tree_first_line: np.ndarray | None = None  # sparse=True: uint64 [num_trees + 1]

    @property
    def sparse(self) -> bool:
        return self.node_lines is not None

    def params(self, sum_mode: int = 0, clusters: int | None = None):
        if self.sparse:
            return make_sparse_params(self.num_trees, self.num_levels, self.num_features, self.missing_bits,
                                      self.cmp_mode, clusters, sum_mode)
        return make_params(self.num_trees, self.num_levels, self.num_features, self.missing_bits, self.cmp_mode,
                           clusters, sum_mode)

    def load_into(self, engine, shard_index: int = 0, shard_count: int = 1, **kw):
        """Load this model into a ddt.Engine; models with classes are scored with Engine.classify*()."""
        if self.num_classes > 1 and "clusters" not in kw:  # every class is its own ensemble of num_trees / num_classes trees
            from .engine import default_clusters
            kw["clusters"] = default_clusters(-(-self.num_trees // self.num_classes))
        if self.sparse:
            return engine.load_model_sparse(self.params(**kw), self.node_lines, self.tree_first_line, shard_index, shard_count,
                                            self.num_classes, True)
        if self.num_classes > 1:
            return engine.load_model_multiclass(self.params(**kw), self.wlines, self.flines, self.num_classes, True, shard_index, shard_count)
        return engine.load_model(self.params(**kw), self.wlines, self.flines, shard_index, shard_count)


LEAF_MIN, LEAF_LIMIT = 2.0 ** -102, 2.0 ** 96   # the loader's exact leaf domain (csrc/ddt_model.cpp leaf_outside_exact_domain)


def leaf_f32(v) -> np.float32:
    """fp32 leaf value inside the engine's exact leaf domain: -0, sub-normals and |v| < 2^-102 become +0; |v| >= 2^96, Inf and
    NaN raise ValueError (see the module docstring)."""
    with np.errstate(over="ignore", under="ignore"):
        f = np.float32(v)
    if not np.isfinite(f) or abs(float(f)) >= LEAF_LIMIT:
        raise ValueError(f"leaf value {v!r} is outside the engine's leaf domain (|v| < 2^96, finite): ddt_load_model* would refuse it")
    if abs(float(f)) < LEAF_MIN:
        return np.float32(0.0)
    return f


def le_to_lt_threshold(t64) -> np.ndarray:
    """fp32 thr such that for every fp32 x:  x <= t64  <=>  x < thr."""
    t64 = np.asarray(t64, np.float64)
    with np.errstate(over="ignore"):
        t32 = t64.astype(np.float32)
    t32 = np.where(t32.astype(np.float64) > t64, np.nextafter(t32, np.float32(-np.inf)), t32)  # largest fp32 <= t64
    return np.nextafter(t32.astype(np.float32), np.float32(np.inf)).astype(np.float32)


class _Tree:
    """Explicit binary tree: arrays indexed by node id; leaf <=> left[n] < 0."""

    def __init__(self, left, right, feature, thr_lt, value, miss_right):
        self.left, self.right, self.feature = np.asarray(left), np.asarray(right), np.asarray(feature)
        self.thr_lt, self.value, self.miss_right = np.asarray(thr_lt, np.float32), np.asarray(value), np.asarray(miss_right)

    def depth(self) -> int:
        d, stack = 0, [(0, 0)]
        while stack:
            n, k = stack.pop()
            if self.left[n] < 0:
                d = max(d, k)
            else:
                stack.append((int(self.left[n]), k + 1))
                stack.append((int(self.right[n]), k + 1))
        return d

    def to_heap(self, D: int, scale: float = 1.0):
        nint, nleaf = (1 << D) - 1, 1 << D
        thr = np.zeros(nint, np.float32)
        fidx = np.zeros(nint, np.uint16)
        mr = np.zeros(nint, np.uint8)
        leaf = np.zeros(nleaf, np.float32)
        stack = [(0, 0, 0, None)]  # (tree node or -1, heap index, depth, frozen leaf value)
        while stack:
            n, h, k, frozen = stack.pop()
            is_leaf = n < 0 or self.left[n] < 0
            if k == D:
                if not is_leaf:
                    raise ValueError("tree deeper than num_levels")
                leaf[h - nint] = leaf_f32((frozen if n < 0 else float(self.value[n])) * scale)
            elif is_leaf:  # pad: dummy node, both children repeat the leaf value
                v = frozen if n < 0 else float(self.value[n])
                stack.append((-1, 2 * h + 1, k + 1, v))
                stack.append((-1, 2 * h + 2, k + 1, v))
            else:
                thr[h], fidx[h], mr[h] = self.thr_lt[n], self.feature[n], self.miss_right[n]
                stack.append((int(self.left[n]), 2 * h + 1, k + 1, None))
                stack.append((int(self.right[n]), 2 * h + 2, k + 1, None))
        return thr, fidx, mr, leaf


    def to_sparse(self, scale: float = 1.0) -> np.ndarray:
        """-> uint32 [n_internal, 4] node lines in breadth-first order (children after their parent)."""
        if self.left[0] < 0:  # a single leaf: one line, both children the value
            v = leaf_f32(float(self.value[0]) * scale).view(np.uint32)
            return np.array([[0, 0xC000, v, v]], np.uint32)
        order, pos = [0], {0: 0}
        for n in order:  # breadth-first over internal nodes
            for c in (int(self.left[n]), int(self.right[n])):
                if self.left[c] >= 0:
                    pos[c] = len(order)
                    order.append(c)
        lines = np.zeros((len(order), 4), np.uint32)
        for k, n in enumerate(order):
            e = int(self.feature[n]) | (int(self.miss_right[n]) << 13)
            lines[k, 0] = np.float32(self.thr_lt[n]).view(np.uint32)
            for side, c in enumerate((int(self.left[n]), int(self.right[n]))):
                if self.left[c] < 0:
                    e |= 1 << (14 + side)
                    lines[k, 2 + side] = leaf_f32(float(self.value[c]) * scale).view(np.uint32)
                else:
                    lines[k, 2 + side] = pos[c]
            lines[k, 1] = e
        return lines


def _pack_sparse(trees, scales, num_features, num_classes=1, base=None, num_levels=None) -> ImportedModel:
    D = max(1, max(t.depth() for t in trees))
    if num_levels is not None:
        if num_levels < D:
            raise ValueError("tree deeper than num_levels")
        D = num_levels
    if D > 64:
        raise ValueError(f"tree depth {D} exceeds the sparse format's 64 levels")
    per = [t.to_sparse(s) for t, s in zip(trees, scales)]
    if any((ln[:, 1] & 0x7FF).max(initial=0) >= num_features for ln in per):
        raise ValueError("feature index out of range")
    first = np.zeros(len(per) + 1, np.uint64)
    first[1:] = np.cumsum([ln.shape[0] for ln in per])
    return ImportedModel(np.zeros(0, np.uint32), np.zeros(0, np.uint16), len(trees), D, num_features, num_classes,
                         np.zeros(num_classes) if base is None else np.asarray(base, np.float64).reshape(-1),
                         node_lines=np.concatenate(per, axis=0), tree_first_line=first)


def _pack(trees, scales, num_features, num_classes=1, base=None, num_levels=None, sparse=False) -> ImportedModel:
    if sparse:
        return _pack_sparse(trees, scales, num_features, num_classes, base, num_levels)
    D = max(1, max(t.depth() for t in trees)) if num_levels is None else num_levels
    if D > MAX_LEVELS:
        raise ValueError(f"tree depth {D} exceeds the format's {MAX_LEVELS} levels")
    T = len(trees)
    wl, fl = weights_lines_per_tree(D) * 4, findex_lines_per_tree(D) * 8
    w = np.zeros((T, wl), np.uint32)
    f = np.zeros((T, fl), np.uint16)
    nint = (1 << D) - 1
    for i, (t, s) in enumerate(zip(trees, scales)):
        thr, fidx, mr, leaf = t.to_heap(D, s)
        if fidx.max(initial=0) >= num_features:
            raise ValueError("feature index out of range")
        w[i, :nint] = thr.view(np.uint32)
        w[i, nint:nint + (1 << D)] = leaf.view(np.uint32)
        f[i, :nint] = fidx | (mr.astype(np.uint16) << 13)
    return ImportedModel(w.reshape(-1), f.reshape(-1), T, D, num_features, num_classes,
                         np.zeros(num_classes) if base is None else np.asarray(base, np.float64).reshape(-1))


# ---- scikit-learn ------------------------------------------------------------------------------------
def _sk_tree(tree_, out_index=0, value_transform=None) -> _Tree:
    t = tree_
    val = t.value[:, out_index, :] if t.value.ndim == 3 else t.value
    val = val[:, 0] if value_transform is None else value_transform(val)
    mgl = getattr(t, "missing_go_to_left", None)
    mr = np.zeros(t.node_count, np.uint8) if mgl is None else (1 - np.asarray(mgl, np.uint8))
    leafmask = t.children_left < 0
    thr = np.where(leafmask, 0.0, t.threshold)
    return _Tree(t.children_left, t.children_right, np.where(leafmask, 0, t.feature), le_to_lt_threshold(thr), val, mr)


def from_sklearn(model, num_levels: int | None = None, sparse: bool = False) -> ImportedModel:
    """score(x) + base_score == model.predict(x) (regressors) / decision_function (boosted classifiers);
    RandomForest/ExtraTrees/DecisionTree classifiers: class score = mean class probability, label = argmax."""
    name = type(model).__name__
    F = int(model.n_features_in_)
    if name in ("DecisionTreeRegressor", "ExtraTreeRegressor"):
        return _pack([_sk_tree(model.tree_)], [1.0], F, num_levels=num_levels, sparse=sparse)
    if name in ("RandomForestRegressor", "ExtraTreesRegressor"):
        n = len(model.estimators_)
        return _pack([_sk_tree(e.tree_) for e in model.estimators_], [1.0 / n] * n, F, num_levels=num_levels, sparse=sparse)
    if name == "GradientBoostingRegressor":
        trees = [_sk_tree(e.tree_) for e in model.estimators_[:, 0]]
        base = float(np.ravel(model.init_.predict(np.zeros((1, F))))[0]) if model.init_ != "zero" else 0.0
        return _pack(trees, [float(model.learning_rate)] * len(trees), F, base=[base], num_levels=num_levels, sparse=sparse)
    if name == "GradientBoostingClassifier":
        K = model.estimators_.shape[1]  # 1 for binary (log-odds of class 1), n_classes otherwise
        trees = [_sk_tree(model.estimators_[s, k].tree_) for s in range(model.estimators_.shape[0]) for k in range(K)]
        raw0 = np.ravel(model._raw_predict_init(np.zeros((1, F))))
        return _pack(trees, [float(model.learning_rate)] * len(trees), F, num_classes=K, base=raw0, num_levels=num_levels, sparse=sparse)
    if name in ("RandomForestClassifier", "ExtraTreesClassifier", "DecisionTreeClassifier", "ExtraTreeClassifier"):
        ests = [model] if name.startswith(("DecisionTree", "ExtraTreeC")) else list(model.estimators_)
        K, n = int(model.n_classes_), len(ests)

        def prob(k):
            return lambda v: v[:, k] / np.maximum(v.sum(axis=1), 1e-300)

        trees = [_sk_tree(e.tree_, 0, prob(k)) for e in ests for k in range(K)]  # interleaved: tree i -> class i % K
        return _pack(trees, [1.0 / n] * len(trees), F, num_classes=K, num_levels=num_levels, sparse=sparse)
    if name in ("HistGradientBoostingRegressor", "HistGradientBoostingClassifier"):
        # model._predictors[iteration][k].nodes: structured array {value, feature_idx, num_threshold, missing_go_to_left,
        # left, right, is_leaf, ...}; a sample goes left iff x <= num_threshold (missing: missing_go_to_left); the
        # learning rate is already folded into the leaf values; raw score = baseline + sum of the leaves.
        K = len(model._predictors[0])
        trees = []
        for it in model._predictors:
            for k in range(K):
                nd = it[k].nodes
                if "is_categorical" in nd.dtype.names and np.any(nd["is_categorical"][~nd["is_leaf"].astype(bool)]):
                    raise TypeError("categorical splits (bitsets) have no counterpart in the reference's threshold format")
                leafmask = nd["is_leaf"].astype(bool)
                left = np.where(leafmask, -1, nd["left"].astype(np.int64))
                right = np.where(leafmask, -1, nd["right"].astype(np.int64))
                thr = np.where(leafmask, 0.0, nd["num_threshold"].astype(np.float64))
                mr = np.where(leafmask, 0, 1 - nd["missing_go_to_left"].astype(np.uint8)).astype(np.uint8)
                trees.append(_Tree(left, right, np.where(leafmask, 0, nd["feature_idx"]), le_to_lt_threshold(thr), nd["value"], mr))
        base = np.ravel(model._baseline_prediction).astype(np.float64)
        return _pack(trees, [1.0] * len(trees), F, num_classes=K, base=base, num_levels=num_levels, sparse=sparse)
    raise TypeError(f"unsupported scikit-learn model {name}")


# ---- XGBoost JSON ---------------------------------------------------------------------------------------
# objective -> how learner_model_param.base_score (stored in the OUTPUT space) becomes the additive margin offset
_XGB_LINK = {
    "reg:squarederror": "identity", "reg:squaredlogerror": "identity", "reg:absoluteerror": "identity",
    "reg:pseudohubererror": "identity", "reg:quantileerror": "identity", "binary:logitraw": "identity",
    "binary:hinge": "identity", "rank:pairwise": "identity", "rank:ndcg": "identity", "rank:map": "identity",
    "multi:softmax": "identity", "multi:softprob": "identity",
    "binary:logistic": "logit", "reg:logistic": "logit",
    "count:poisson": "log", "reg:gamma": "log", "reg:tweedie": "log", "survival:cox": "log",
}


def _xgb_base_scores(lmp, objective: str, K: int) -> np.ndarray:
    """Margin-space offset(s).  base_score comes as "0.5", as a bracketed string "[5E-1]" / "[1E0,2E0]" (XGBoost >= 2,
    which also estimates it from the data) or as a list."""
    raw = lmp.get("base_score", "0.5")
    if isinstance(raw, (list, tuple)):
        vals = [float(v) for v in raw]
    else:
        txt = str(raw).strip()
        vals = [float(v) for v in txt.strip("[]").split(",") if v.strip()] if txt.startswith("[") else [float(txt)]
    if not vals:
        vals = [0.5]
    link = _XGB_LINK.get(objective)
    if link is None:
        raise TypeError(f"unsupported XGBoost objective {objective!r}: the link of base_score is not known")
    v = np.asarray(vals, np.float64)
    if link == "logit":
        if np.any((v <= 0) | (v >= 1)):
            raise ValueError("base_score of a logistic objective must lie in (0, 1)")
        v = np.log(v / (1.0 - v))
    elif link == "log":
        if np.any(v <= 0):
            raise ValueError("base_score of a log-link objective must be positive")
        v = np.log(v)
    if v.size == 1:
        v = np.repeat(v, K)
    if v.size != K:
        raise ValueError(f"{v.size} base scores for {K} classes")
    return v


def from_xgboost_json(path_or_dict, num_levels: int | None = None, sparse: bool = False) -> ImportedModel:
    """XGBoost `save_model("*.json")` (gbtree).  XGBoost goes left iff x < split_condition -- the reference's rule;
    `default_left` -> miss_right = 0.  score + base_score == margin, base_score being the model's
    learner_model_param.base_score mapped through the objective's link (logit for binary:logistic / reg:logistic, log
    for poisson / gamma / tweedie, identity otherwise; unknown objectives are refused); multi-class: tree_info gives
    the class."""
    j = path_or_dict if isinstance(path_or_dict, dict) else json.load(open(path_or_dict))
    learner = j["learner"]
    gb = learner["gradient_booster"]
    if gb.get("name", "gbtree") not in ("gbtree",):
        raise TypeError(f"unsupported booster {gb.get('name')}")
    model = gb["model"]
    lmp = learner["learner_model_param"]
    F = int(lmp["num_feature"])
    K = max(1, int(lmp.get("num_class", "0")))
    objective = learner.get("objective", {}).get("name", "reg:squarederror")
    base = _xgb_base_scores(lmp, objective, K)
    info = [int(v) for v in model.get("tree_info", [0] * len(model["trees"]))]
    trees = []
    for t in model["trees"]:
        left = np.asarray(t["left_children"], np.int64)
        right = np.asarray(t["right_children"], np.int64)
        cond = np.asarray(t["split_conditions"], np.float32)
        feat = np.asarray(t["split_indices"], np.int64)
        dl = np.asarray(t["default_left"], np.uint8)
        leafmask = left < 0
        trees.append(_Tree(left, right, np.where(leafmask, 0, feat), np.where(leafmask, np.float32(0), cond), cond, 1 - dl))
    if K > 1:  # our class rule is interleaved i % K; XGBoost's tree_info normally is exactly that
        if any(c != i % K for i, c in enumerate(info)):
            raise ValueError("tree_info is not round-robin over classes")
    return _pack(trees, [1.0] * len(trees), F, num_classes=K, base=base, num_levels=num_levels, sparse=sparse)
