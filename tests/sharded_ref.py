"""TEST INFRASTRUCTURE (moved out of the package in round 3): a Python mirror of the multi-GPU chunk pipeline, driven through
torch.distributed.  The PRODUCT pipeline is C++ behind the C-ABI (csrc/ddt_comm.cpp: ddt_comm_* / ddt_score_sharded_device); this
mirror exists because RCCL cannot put several ranks on one GPU and has no CPU backend: the gloo world-size-2/4 CPU tests
(tests/test_sharded_gloo.py), the two-ranks-on-one-GPU tests and `bench.py --backend gloo` run the same schedule through it.

Tree-sharded multi-GPU scoring: one process per GPU, torch.distributed ("nccl" == RCCL over xGMI).

Reference mode reproduced (rtl/DTEngine/DTInference.sv:28-37, mode "trees partitioned, tuples broadcast,
results aggregated"): rank g holds the contiguous tree shard g of ceil(T/G) trees
(PCIeReceiver.sv:241-264), every rank scores ALL tuples against its shard, and the per-tuple partial
scores are summed across ranks -- the reference does that with a chain of fp32 adders, host -> dev1 ->
... (ResultsCombiner.sv:292-311,359-369).  Two combine modes:

  "allreduce"  one RCCL all-reduce (sum) per chunk of partial scores -- the collective BASELINE.json
               names.  The ring's summation order is RCCL's, so results match the reference chain order
               to fp32 rounding (|err| <= ~G ulp), not bit for bit.
  "chain"      deterministic: all-to-all (each rank receives every rank's slice of its 1/G segment),
               fixed-order chain add p0+p1+...+p(G-1) on the owner, all-gather.  Same bytes on the wire
               as a ring all-reduce, point-to-point over the fully connected xGMI mesh, and bit-exact
               with the reference's chain order.

Chunks are pipelined: chunk k's collective runs on RCCL's stream while chunk k+1 is being scored.
"""
from __future__ import annotations

from typing import Callable, Optional


from ddt.engine import shard_bounds  # noqa: E402,F401  (the split itself is the C-ABI's: ddt_load_model_shard)


class RowShardedScorer:
    """The reference's OTHER multi-device mode (rtl/DTEngine/DTInference.sv:28-37, PCIeReceiver.sv:289-312): every
    device holds the whole ensemble, the tuples are partitioned, results are interleaved -- "replicas only", no
    arithmetic crosses devices.  Rank r scores rows [r*ceil(n/G), ...) and an all-gather gives every rank the full
    score vector (skip it with gather=False to leave the scores sharded).  Each engine must hold ALL trees."""

    def __init__(self, engine, group=None, gather: bool = True):
        import torch.distributed as dist

        self.engine, self.group, self.gather = engine, group, gather
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"

    def score(self, tuples, out=None):
        import torch
        import torch.distributed as dist

        from ddt.engine import tuple_words

        W = tuple_words(self.engine.params.num_features)
        n = tuples.numel() // W
        tuples = tuples.reshape(n, W)
        per = (n + self.G - 1) // self.G
        lo, hi = min(self.rank * per, n), min((self.rank + 1) * per, n)
        full = torch.zeros(self.G * per, dtype=torch.float32, device=tuples.device)
        mine = full[self.rank * per: self.rank * per + per]
        if hi > lo:
            self.engine.score_device(tuples[lo:hi], out=mine[: hi - lo])
        if self.G > 1 and self.gather:
            if self.gloo and full.is_cuda:
                h = torch.empty(self.G * per, dtype=torch.float32)
                dist.all_gather_into_tensor(h, mine.cpu().contiguous(), group=self.group)
                full.copy_(h)
            else:
                dist.all_gather_into_tensor(full, mine.clone(), group=self.group)
        res = full[:n]
        if out is not None:
            out.copy_(res)
            return out
        return res


def chain_sum(parts):
    """parts [G, n] (torch, any device) -> (((p0 + p1) + p2) + ...), fp32, the reference's hop order."""
    run = parts[0].clone()
    for g in range(1, parts.shape[0]):
        run = parts[g] + run  # local + upstream, ResultsCombiner.sv:292-311
    return run


class ShardedScorer:
    """Combine per-rank partial scores into full scores on every rank.

    partial_fn(tuples, out) must write this rank's partial scores for `tuples` into `out` (asynchronously
    on the current stream for CUDA tensors).  chain_fn(parts[G, n]) -> [n] is the ordered chain add
    (defaults to the torch implementation above; the GPU path passes Engine.chain_sum_device).
    """

    def __init__(self, partial_fn: Callable, tuple_words: int, group=None, mode: str = "allreduce",
                 chunk_rows: int = 1 << 23, chain_fn: Optional[Callable] = None, force_collectives: bool = False):
        import torch.distributed as dist

        assert mode in ("allreduce", "chain")
        self.partial_fn, self.W, self.group, self.mode = partial_fn, tuple_words, group, mode
        self.force = bool(force_collectives)  # run the chunk pipeline + collectives even in a one-rank group (sanity / overhead runs)
        self.chunk_rows = int(chunk_rows)
        self.chain_fn = chain_fn or chain_sum
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # gloo has no device collectives: device tensors are staged through host memory (functional test mode
        # for boxes with one GPU; the production path is backend "nccl" == RCCL over xGMI)
        self.gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"

    @classmethod
    def from_engine(cls, engine, **kw):
        from ddt.engine import tuple_words

        def partial(tuples, out):
            engine.score_device(tuples, out=out)

        def chain(parts):
            return engine.chain_sum_device(parts.contiguous())

        return cls(partial, tuple_words(engine.params.num_features), chain_fn=chain, **kw)

    def score(self, tuples, out=None):
        """tuples: [n, W] 4-byte tensor (replicated on every rank) -> fp32 [n] full scores on every rank."""
        import torch
        import torch.distributed as dist

        n = tuples.numel() // self.W
        tuples = tuples.reshape(n, self.W)
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=tuples.device)
        if self.G == 1 and not self.force:
            self.partial_fn(tuples, out)
            return out
        works, keep = [], []
        G = self.G
        for lo in range(0, n, self.chunk_rows):
            hi = min(n, lo + self.chunk_rows)
            if self.mode == "allreduce":
                o = out[lo:hi]
                self.partial_fn(tuples[lo:hi], o)
                if self.gloo and o.is_cuda:
                    h = o.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                    o.copy_(h)
                else:
                    works.append(dist.all_reduce(o, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            elif self.gloo and tuples.is_cuda:
                m = hi - lo
                seg = (m + G - 1) // G
                part = torch.zeros(G * seg, dtype=torch.float32, device=tuples.device)
                self.partial_fn(tuples[lo:hi], part[:m])
                recv = torch.empty(G * seg, dtype=torch.float32)
                dist.all_to_all_single(recv, part.cpu(), group=self.group)
                mine = self.chain_fn(recv.to(tuples.device).view(G, seg))
                full = torch.empty(G * seg, dtype=torch.float32)
                dist.all_gather_into_tensor(full, mine.cpu().contiguous(), group=self.group)
                out[lo:hi] = full[:m].to(tuples.device)
            else:
                m = hi - lo
                seg = (m + G - 1) // G  # segment owned by each rank (last one zero padded)
                part = torch.zeros(G * seg, dtype=torch.float32, device=tuples.device)
                self.partial_fn(tuples[lo:hi], part[:m])
                recv = torch.empty(G * seg, dtype=torch.float32, device=tuples.device)
                dist.all_to_all_single(recv, part, group=self.group)  # recv[g*seg:(g+1)*seg] = rank g's slice of my segment
                mine = self.chain_fn(recv.view(G, seg))
                full = torch.empty(G * seg, dtype=torch.float32, device=tuples.device)
                works.append(dist.all_gather_into_tensor(full, mine.contiguous(), group=self.group, async_op=True))
                keep.append((lo, hi, full, part, recv, mine))
        for w in works:
            w.wait()
        for (lo, hi, full, *_rest) in keep:
            out[lo:hi] = full[: hi - lo]
        return out


class ShardedClassifier:
    """BASELINE config 5 across GPUs: every rank holds shard g of EVERY class (ddt_load_model_multiclass with
    shard_index / shard_count), scores all tuples, the per-class partial sums [K, n] are combined across ranks exactly
    like the scalar scores of ShardedScorer (an all-reduce, or the deterministic chain), and the argmax runs on the
    combined sums.  partial_fn(tuples, out[K, m]) writes this rank's partial class scores; argmax_fn([K, n]) -> int32 [n].
    """

    def __init__(self, partial_fn: Callable, tuple_words: int, num_classes: int, argmax_fn: Callable, group=None,
                 mode: str = "allreduce", chunk_rows: int = 1 << 22, chain_fn: Optional[Callable] = None):
        import torch.distributed as dist

        assert mode in ("allreduce", "chain")
        self.partial_fn, self.W, self.K, self.argmax_fn = partial_fn, tuple_words, int(num_classes), argmax_fn
        self.group, self.mode, self.chunk_rows = group, mode, int(chunk_rows)
        self.chain_fn = chain_fn or chain_sum
        self.G = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.gloo = dist.is_initialized() and dist.get_backend(group) == "gloo"

    @classmethod
    def from_engine(cls, engine, **kw):
        from ddt.engine import tuple_words

        def partial(tuples, out):
            engine.classify_device(tuples, class_scores=out, want_labels=False)

        def chain(parts):
            return engine.chain_sum_device(parts.contiguous())

        return cls(partial, tuple_words(engine.params.num_features), engine.num_classes, engine.argmax_device,
                   chain_fn=chain, **kw)

    def _combine(self, pc):
        """pc: contiguous [K, m] partial sums of this rank -> (combined [K, m] tensor, async work or None)."""
        import torch
        import torch.distributed as dist

        G = self.G
        if G == 1:
            return pc, None
        stage = self.gloo and pc.is_cuda  # gloo has no device collectives: stage through host memory (functional mode)
        if self.mode == "allreduce":
            if stage:
                h = pc.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                pc.copy_(h)
                return pc, None
            return pc, dist.all_reduce(pc, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        flat = pc.reshape(-1)
        m = flat.numel()
        seg = (m + G - 1) // G
        dev = torch.device("cpu") if stage else pc.device
        part = torch.zeros(G * seg, dtype=torch.float32, device=dev)
        part[:m] = flat.to(dev)
        recv = torch.empty(G * seg, dtype=torch.float32, device=dev)
        dist.all_to_all_single(recv, part, group=self.group)
        mine = self.chain_fn(recv.to(pc.device).view(G, seg)).to(dev).contiguous()
        full = torch.empty(G * seg, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(full, mine, group=self.group)
        return full[:m].to(pc.device).view(pc.shape), None

    def classify(self, tuples):
        """tuples [n, W] (replicated on every rank) -> (labels int32 [n], class scores fp32 [K, n]) on every rank."""
        import torch

        n = tuples.numel() // self.W
        tuples = tuples.reshape(n, self.W)
        scores = torch.empty((self.K, n), dtype=torch.float32, device=tuples.device)
        pending = []
        for lo in range(0, n, self.chunk_rows):
            hi = min(n, lo + self.chunk_rows)
            pc = torch.empty((self.K, hi - lo), dtype=torch.float32, device=tuples.device)
            self.partial_fn(tuples[lo:hi], pc)
            pending.append((lo, hi) + self._combine(pc))
        for lo, hi, comb, work in pending:
            if work is not None:
                work.wait()
            scores[:, lo:hi] = comb
        return self.argmax_fn(scores), scores
