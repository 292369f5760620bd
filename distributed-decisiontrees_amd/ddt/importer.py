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
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field

import numpy as np

from .engine import MISSING_DEFAULT, findex_lines_per_tree, make_params, weights_lines_per_tree

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

    def params(self, sum_mode: int = 0, clusters: int | None = None):
        return make_params(self.num_trees, self.num_levels, self.num_features, self.missing_bits, self.cmp_mode,
                           clusters, sum_mode)


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
                leaf[h - nint] = np.float32((frozen if n < 0 else float(self.value[n])) * scale)
            elif is_leaf:  # pad: dummy node, both children repeat the leaf value
                v = frozen if n < 0 else float(self.value[n])
                stack.append((-1, 2 * h + 1, k + 1, v))
                stack.append((-1, 2 * h + 2, k + 1, v))
            else:
                thr[h], fidx[h], mr[h] = self.thr_lt[n], self.feature[n], self.miss_right[n]
                stack.append((int(self.left[n]), 2 * h + 1, k + 1, None))
                stack.append((int(self.right[n]), 2 * h + 2, k + 1, None))
        return thr, fidx, mr, leaf


def _pack(trees, scales, num_features, num_classes=1, base=None, num_levels=None) -> ImportedModel:
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


def from_sklearn(model, num_levels: int | None = None) -> ImportedModel:
    """score(x) + base_score == model.predict(x) (regressors) / decision_function (boosted classifiers);
    RandomForest/ExtraTrees/DecisionTree classifiers: class score = mean class probability, label = argmax."""
    name = type(model).__name__
    F = int(model.n_features_in_)
    if name in ("DecisionTreeRegressor", "ExtraTreeRegressor"):
        return _pack([_sk_tree(model.tree_)], [1.0], F, num_levels=num_levels)
    if name in ("RandomForestRegressor", "ExtraTreesRegressor"):
        n = len(model.estimators_)
        return _pack([_sk_tree(e.tree_) for e in model.estimators_], [1.0 / n] * n, F, num_levels=num_levels)
    if name == "GradientBoostingRegressor":
        trees = [_sk_tree(e.tree_) for e in model.estimators_[:, 0]]
        base = float(np.ravel(model.init_.predict(np.zeros((1, F))))[0]) if model.init_ != "zero" else 0.0
        return _pack(trees, [float(model.learning_rate)] * len(trees), F, base=[base], num_levels=num_levels)
    if name == "GradientBoostingClassifier":
        K = model.estimators_.shape[1]  # 1 for binary (log-odds of class 1), n_classes otherwise
        trees = [_sk_tree(model.estimators_[s, k].tree_) for s in range(model.estimators_.shape[0]) for k in range(K)]
        raw0 = np.ravel(model._raw_predict_init(np.zeros((1, F))))
        return _pack(trees, [float(model.learning_rate)] * len(trees), F, num_classes=K, base=raw0, num_levels=num_levels)
    if name in ("RandomForestClassifier", "ExtraTreesClassifier", "DecisionTreeClassifier", "ExtraTreeClassifier"):
        ests = [model] if name.startswith(("DecisionTree", "ExtraTreeC")) else list(model.estimators_)
        K, n = int(model.n_classes_), len(ests)

        def prob(k):
            return lambda v: v[:, k] / np.maximum(v.sum(axis=1), 1e-300)

        trees = [_sk_tree(e.tree_, 0, prob(k)) for e in ests for k in range(K)]  # interleaved: tree i -> class i % K
        return _pack(trees, [1.0 / n] * len(trees), F, num_classes=K, num_levels=num_levels)
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
        return _pack(trees, [1.0] * len(trees), F, num_classes=K, base=base, num_levels=num_levels)
    raise TypeError(f"unsupported scikit-learn model {name}")


# ---- XGBoost JSON ---------------------------------------------------------------------------------------
def from_xgboost_json(path_or_dict, num_levels: int | None = None) -> ImportedModel:
    """XGBoost `save_model("*.json")` (gbtree).  XGBoost goes left iff x < split_condition -- the reference's rule;
    `default_left` -> miss_right = 0.  score + base_score == margin; multi-class: tree_info gives the class."""
    j = path_or_dict if isinstance(path_or_dict, dict) else json.load(open(path_or_dict))
    learner = j["learner"]
    gb = learner["gradient_booster"]
    if gb.get("name", "gbtree") not in ("gbtree",):
        raise TypeError(f"unsupported booster {gb.get('name')}")
    model = gb["model"]
    lmp = learner["learner_model_param"]
    F = int(lmp["num_feature"])
    K = max(1, int(lmp.get("num_class", "0")))
    base = float(lmp.get("base_score", "0.5")) if not isinstance(lmp.get("base_score"), list) else float(lmp["base_score"][0])
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
    return _pack(trees, [1.0] * len(trees), F, num_classes=K, base=[base] * K, num_levels=num_levels)
