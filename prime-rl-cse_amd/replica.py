"""Data-parallel training of a model that fits whole on every MI355X.

The reference shards every model with FSDP2 and, when `model.compile` is set,
hands the blocks to a tracing compiler (src/prime_rl/trainer/model.py:102-135,
:219-230).  For the models of its smaller recipes (0.6B .. 7B parameters) a
288 GB GPU does not need the parameters sharded, and what limits the step is the
HOST: a micro-batch is ~2500 kernel launches, and the GPU waits for Python a
quarter of the time.  So here:

  * every rank holds the whole model in bf16 in ONE flat buffer (the module's
    parameters are views into it), and the whole fp32 gradient sum in another;
  * the fp32 master weights and the optimizer state are sharded: rank r owns
    elements [r * n / W, (r + 1) * n / W) of the flat buffer (what FSDP2's fp32
    shards plus `MixedPrecisionPolicy(param_dtype=bf16, reduce_dtype=fp32)` are
    in the reference);
  * a micro-batch (forward, the HIP loss path, backward, gradient accumulation)
    runs eagerly by default.  With `graphs=True` it is captured once per input
    shape in a HIP graph and replayed (one launch from the host instead of
    thousands), but only if the captured graph holds no memset node: library
    calls record them under capture, replays of such nodes have gone wrong on
    MI355X boxes, and a graph that holds one is refused (`GraphHoldsMemsets`).  Shapes seen for the first time, and
    more distinct shapes than `max_graphs`, run eagerly through the same code
    (a graph keeps its activations' memory: ~30 GB for a 16 K-token row of the
    0.6B model, so `max_graphs` = 4 is sized for 288 GB);
  * a step ends with ONE fp32 reduce-scatter (mean over ranks), the gradient
    norm, the optimizer on the shard and ONE bf16 all-gather: the same traffic
    as FSDP2 with parameters kept gathered, issued outside the graphs.

The numbers that depend on the step (the loss normaliser) are read from device
memory by the captured kernels, never baked into a graph.
"""

from __future__ import annotations

import re
import time
from collections import OrderedDict
from typing import Any

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor, nn

from . import _capi as C
from .batch import host_to_device, to_device
from .config import LossConfig
from .data import MicroBatch
from .loss import policy_loss, unit_gradient
from .utils import Tensors

ALIGN = 128  # elements: every parameter starts on a 256-byte boundary of the bf16 buffer


def _round_up(n: int, a: int) -> int:
    return (n + a - 1) // a * a


_LAYER = re.compile(r"^((?:.*\.)?layers\.\d+)\.")  # parameters of one transformer block share this prefix


def bucket_plan(names: list[str], numel: list[int], bucket_elems: int | None) -> list[tuple[int, int]]:
    """Consecutive parameters [first, last) per exchange bucket.  A transformer block is never cut (its
    parameters get their gradients within a few kernels of each other, and the forward needs all of them at
    once); blocks and the parameters outside them (embedding table, final norm, an untied LM head) are joined
    front to back while a bucket stays within `bucket_elems` elements (a group of less than 1/64 of that joins
    the bucket before it regardless).  None: one bucket for everything."""
    if not names:
        return []
    if bucket_elems is None:
        return [(0, len(names))]
    groups: list[list] = []  # [key, first, last, elements]
    for i, (name, n) in enumerate(zip(names, numel)):
        m = _LAYER.match(name)
        key = m.group(1) if m else name
        if groups and groups[-1][0] == key:
            groups[-1][2] = i + 1
            groups[-1][3] += n
        else:
            groups.append([key, i, i + 1, n])
    out: list[list[int]] = []
    for _, first, last, n in groups:
        if out and (out[-1][2] + n <= bucket_elems or n * 64 <= bucket_elems):  # (small fry never opens a bucket)
            out[-1][1] = last
            out[-1][2] += n
        else:
            out.append([first, last, n])
    return [(a, b) for a, b, _ in out]


class Bucket:
    """Parameters [first, last) of the flat buffers, elements [lo, hi); of these every rank owns `shard`
    consecutive elements, rank r those from lo + r * shard, kept at [shard_lo, shard_lo + shard) of its master
    shard.  One reduce-scatter of the gradient sums and one all-gather of the new compute copy per bucket
    and step."""

    __slots__ = ("index", "first", "last", "lo", "hi", "shard", "shard_lo", "reduce_work", "gather_work", "reduced",
                 "held", "pending", "launched", "chunk_lo", "chunk_hi")

    def __init__(self, index: int, first: int, last: int):
        self.index, self.first, self.last = index, first, last
        self.lo = self.hi = self.shard = self.shard_lo = 0
        self.reduce_work: Any = None  # the reduce-scatter in flight (RCCL), until somebody has waited for it
        self.gather_work: Any = None  # the all-gather in flight
        self.reduced = False          # this step's gradient sums of the bucket have been handed to the exchange
        self.held = False             # Replica: reduced only after the pass (the embedding table's gradient comes last)
        self.launched = False
        self.chunk_lo = self.chunk_hi = 0


class FlatState:
    """The flat buffers of one rank and the end of a step over them (no HIP code in here: plain torch and
    torch.distributed, so it also runs on the CPU over gloo).

    `flat16`  [n]      compute copy of all parameters (the model's parameters are views into it)
    `acc32`   [n]      this rank's gradient sum over the micro-batches of the step, fp32
    `master`  [n / W]  this rank's shard of the fp32 master weights (an nn.Parameter: the optimizer's only one)

    The buffers are cut into buckets (`bucket_plan`).  Every bucket is sharded over the ranks by itself, so that
    a bucket's gradient sums can be reduce-scattered as soon as backward has produced them, while backward goes on
    (the reference gets the same overlap from FSDP2's per-block groups, src/prime_rl/trainer/model.py:102-135),
    and its new weights can be all-gathered while the next forward is already running on earlier buckets.  The
    master shard is the concatenation of the rank's parts of all buckets.  With one bucket (the default) the
    layout is "rank r owns elements [r * n / W, (r + 1) * n / W)".  AdamW and SGD are element-wise, and every
    element's gradient is the sum over ranks whatever bucket it is in, so bucketing changes no value beyond the
    order in which the collective adds the ranks' contributions (one order only with two ranks).
    """

    def __init__(self, model: nn.Module, group: Any = None, compute_dtype: torch.dtype = torch.bfloat16,
                 bucket_bytes: int | None = None, exchange_alone: bool = False):
        """`bucket_bytes`: target size of a bucket's fp32 gradient sums (None: one bucket).  `exchange_alone`:
        a single rank still goes through the collectives (over its one-member group), for tests of that path."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.exchange = self.world > 1 or (exchange_alone and dist.is_initialized())
        named = list(model.named_parameters())
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        if not self.params:
            raise ValueError("FlatState: the model has no parameters")
        dev = self.params[0].device
        self.numel = [p.numel() for p in self.params]
        plan = bucket_plan(self.names, self.numel, None if bucket_bytes is None else max(int(bucket_bytes) // 4, 1))
        self.buckets = [Bucket(k, a, b) for k, (a, b) in enumerate(plan)]
        self.bucket_of = [0] * len(self.params)
        self.offsets, off, shard_lo = [], 0, 0
        for b in self.buckets:
            b.lo = off
            for i in range(b.first, b.last):
                self.bucket_of[i] = b.index
                self.offsets.append(off)
                off += _round_up(self.numel[i], ALIGN)
            off = b.lo + _round_up(off - b.lo, ALIGN * self.world)
            b.hi = off
            b.shard = (b.hi - b.lo) // self.world
            b.shard_lo = shard_lo
            shard_lo += b.shard
        self.total = off
        self.shard = self.total // self.world
        self.flat16 = torch.zeros(self.total, dtype=compute_dtype, device=dev)
        master = torch.zeros(self.shard, dtype=torch.float32, device=dev)
        for i, (p, o, n) in enumerate(zip(self.params, self.offsets, self.numel)):
            src = p.detach().reshape(-1)
            self.flat16[o : o + n].copy_(src)
            b = self.buckets[self.bucket_of[i]]
            lo = b.lo + self.rank * b.shard  # this rank's part of the bucket, in flat coordinates
            a, z = max(o, lo), min(o + n, lo + b.shard)
            if a < z:
                master[b.shard_lo + a - lo : b.shard_lo + z - lo].copy_(src[a - o : z - o])
            p.data = self.flat16[o : o + n].view(p.shape)  # the fp32 original is released here
        self.master = nn.Parameter(master)
        self.acc32 = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self._shard_grad = torch.zeros(self.shard, dtype=torch.float32, device=dev) if self.exchange else None
        self._shard16 = torch.zeros(self.shard, dtype=compute_dtype, device=dev) if self.exchange else None

    def holder(self) -> nn.Module:
        """A module whose only parameter is the master shard: what the optimizer is built over.  AdamW and SGD
        are element-wise, so one flat parameter takes exactly the steps the separate parameters would."""
        return nn.ParameterList([self.master])

    def owned_ranges(self) -> list[tuple[int, int, int]]:
        """(flat_lo, flat_hi, master_lo) of every stretch of the flat buffers whose master weights this rank holds."""
        return [(b.lo + self.rank * b.shard, b.lo + (self.rank + 1) * b.shard, b.shard_lo) for b in self.buckets]

    def _gloo(self) -> bool:
        return dist.is_initialized() and dist.get_backend(self.group) == "gloo"

    # ------------------------------------------------------------------ the exchange, bucket by bucket

    def reduce_bucket(self, b: Bucket) -> None:
        """Mean over the ranks of bucket `b`'s gradient sums, this rank's part of it into its gradient shard.
        Over RCCL the reduce-scatter is started behind what the current stream holds so far and left in
        flight (`wait_reduced` makes the current stream wait for it); over gloo it is done on return."""
        if not self.exchange or b.reduced:
            return
        b.reduced = True
        grad = self._shard_grad[b.shard_lo : b.shard_lo + b.shard]
        sums = self.acc32[b.lo : b.hi]
        if self._gloo():  # no reduce-scatter in gloo, host tensors only: reduce the bucket, keep the own part
            total = sums.cpu()
            dist.all_reduce(total, group=self.group)
            grad.copy_(total[self.rank * b.shard : (self.rank + 1) * b.shard]).div_(self.world)
        else:
            b.reduce_work = dist.reduce_scatter_tensor(grad, sums, op=dist.ReduceOp.AVG, group=self.group,
                                                       async_op=True)

    def wait_reduced(self) -> None:
        """Every bucket reduced (those nobody has started are started here) and the current stream behind all of it."""
        for b in self.buckets:
            self.reduce_bucket(b)
        for b in self.buckets:
            if b.reduce_work is not None:
                b.reduce_work.wait()
                b.reduce_work = None

    def _gather_compute_copy(self) -> None:
        """Every rank's new part of every bucket of the compute copy into every rank's `flat16`: one all-gather
        per bucket, front to back, over RCCL left in flight (`wait_gathered`)."""
        for b in self.buckets:
            part = self._shard16[b.shard_lo : b.shard_lo + b.shard]
            if self._gloo():  # gloo gets host tensors only
                parts = [torch.empty(b.shard, dtype=self.flat16.dtype) for _ in range(self.world)]
                dist.all_gather(parts, part.cpu(), group=self.group)
                self.flat16[b.lo : b.hi].copy_(torch.cat(parts))
            else:
                b.gather_work = dist.all_gather_into_tensor(self.flat16[b.lo : b.hi], part, group=self.group,
                                                            async_op=True)

    def wait_gathered(self, b: Bucket | None = None) -> None:
        """The current stream waits until bucket `b` (None: every bucket) of `flat16` holds the new weights."""
        for x in (self.buckets if b is None else (b,)):
            if x.gather_work is not None:
                x.gather_work.wait()
                x.gather_work = None

    def finish_step(self, optimizer: torch.optim.Optimizer, max_norm: float | None, keep_sums: bool = False,
                    gathers_in_flight: bool = False, squared_norm: Tensor | None = None) -> Tensor:
        """Mean of the ranks' gradient sums, clip by the global norm, optimizer step on the shard, new compute
        copy on every rank.  Returns the gradient norm before clipping (a 0-d tensor on the device; reading it
        is the caller's one host synchronisation).  Same clipping rule as torch.nn.utils.clip_grad_norm_,
        which the reference calls (src/prime_rl/trainer/rl/train.py:336-338).  `keep_sums` leaves the gradient
        sums as they are (for a caller whose next step overwrites them); otherwise they are cleared.  Buckets
        whose reduction a caller has already started (`reduce_bucket`) are only waited for.  With
        `gathers_in_flight` the all-gathers of the new compute copy (RCCL) are left running and it is the caller
        who makes a stream wait for a bucket before reading it (`wait_gathered`).  `squared_norm` (one rank only): the
        squared norm of the gradient sums as a 0-d fp32 tensor, from a caller who has it already."""
        if squared_norm is not None and self.exchange:
            raise ValueError("finish_step: `squared_norm` is for a rank that exchanges nothing")
        try:
            return self._finish_step(optimizer, max_norm, keep_sums, squared_norm)
        finally:
            if not gathers_in_flight:
                self.wait_gathered()

    def _finish_step(self, optimizer: torch.optim.Optimizer, max_norm: float | None, keep_sums: bool,
                     squared_norm: Tensor | None = None) -> Tensor:
        self.wait_gathered()  # (a step without any forward: the last step's weights must have arrived)
        if self.exchange:
            self.wait_reduced()
            grad = self._shard_grad
        else:
            grad = self.acc32
        if squared_norm is not None:
            sq = squared_norm
        else:
            sq = torch.linalg.vector_norm(grad, 2, dtype=torch.float32).square()
        if self.exchange:
            if self._gloo():
                host = sq.cpu()
                dist.all_reduce(host, group=self.group)
                sq = host.to(sq.device)
            else:
                dist.all_reduce(sq, group=self.group)
        norm = sq.sqrt()
        coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0).reshape(1) if max_norm is not None else None
        for b in self.buckets:
            b.reduced = False
        if hasattr(optimizer, "step_flat"):
            # clipping, the update, the new compute copy and (one rank) the clearing of the sums in one launch
            if self.exchange:
                optimizer.step_flat(grad, coef, self._shard16)
                self._gather_compute_copy()
                if not keep_sums:
                    self.acc32.zero_()
            else:
                optimizer.step_flat(grad, coef, self.flat16, zero_grad=not keep_sums)
            return norm
        if coef is not None:
            grad.mul_(coef)
        self.master.grad = grad
        optimizer.step()
        self.master.grad = None
        if self.exchange:
            self._shard16.copy_(self.master.detach())
            self._gather_compute_copy()
        else:
            self.flat16.copy_(self.master.detach())
        if not keep_sums:
            self.acc32.zero_()
        return norm


def ctypes_ptr(t: Tensor, byte_offset: int = 0) -> Any:
    import ctypes

    return ctypes.c_void_p(t.data_ptr() + byte_offset)


class GraphHoldsMemsets(RuntimeError):
    """A captured pass holds memset nodes.  Replays of such graphs have gone wrong on MI355X boxes (a node left
    the memory it should clear as it was), so the graph is thrown away before it is ever launched."""


def graph_node_counts(graph: Any) -> tuple[int, int]:
    """(nodes, memset nodes) of a captured `torch.cuda.CUDAGraph(keep_graph=True)`; host-side queries only."""
    import ctypes

    nodes, memsets = ctypes.c_int64(0), ctypes.c_int64(0)
    C.check(C.load().prl_graph_node_counts(ctypes.c_void_p(int(graph.raw_cuda_graph())), ctypes.byref(nodes),
                                           ctypes.byref(memsets)), "prl_graph_node_counts")
    return int(nodes.value), int(memsets.value)


class _Captured:
    """One captured micro-batch shape: the graph, the buffers it reads and the tensors it leaves behind."""

    def __init__(self) -> None:
        self.graph: Any = None
        self.inputs: dict[str, Tensor] = {}
        self.plan: Any = None
        self.loss: Tensor | None = None
        self.aux: dict[str, Tensor] = {}
        self.src: Tensor | None = None
        self.embed_grad: Tensor | None = None  # gradient of the looked-up embedding rows, [1, T, H]


_TENSOR_KEYS = ("input_ids", "position_ids", "advantages", "logprobs", "loss_mask")
_AUX_KEYS = ("logprobs", "entropy", "importance_ratio", "clipped_importance_ratio", "is_clipped", "n_seg")


class ExchangeOrder:
    """The order in which a rank hands its buckets to the gradient exchange.  Collectives must be issued in ONE
    order on every rank, whatever order a rank's gradients land in and however many passes it runs: every
    bucket that is not `held` goes back to front (a bucket only after all behind it), the held ones (the
    embedding table's, whose second share exists only after the pass) front to back at the end.  `landed`
    releases buckets from inside backward as far as that order allows; `finish` names what is left; a rank
    without any pass issues `sequence()` as it stands."""

    def __init__(self, held: list[bool]):
        self.held = list(held)
        self.pending = [0] * len(held)
        self.launched = [False] * len(held)
        self.next = -1

    def sequence(self) -> list[int]:
        n = len(self.held)
        return [k for k in reversed(range(n)) if not self.held[k]] + [k for k in range(n) if self.held[k]]

    def begin(self, pending: list[int]) -> None:
        """`pending[k]`: how many gradients of bucket k this pass will deliver."""
        self.pending = list(pending)
        self.launched = [False] * len(self.held)
        self.next = len(self.held) - 1

    def landed(self, k: int) -> list[int]:
        """A gradient of bucket k has arrived: the buckets to start now, in order."""
        self.pending[k] -= 1
        out = []
        while self.next >= 0:
            if self.held[self.next]:
                self.next -= 1
                continue
            if self.pending[self.next] > 0:
                break
            out.append(self.next)
            self.launched[self.next] = True
            self.next -= 1
        return out

    def finish(self) -> tuple[list[int], list[int]]:
        """After the pass: (buckets not yet started, back to front: some parameter got no gradient; the held
        buckets, front to back)."""
        n = len(self.held)
        rest = [k for k in reversed(range(n)) if not self.launched[k] and not self.held[k]]
        for k in rest:
            self.launched[k] = True
        self.next = -1
        return rest, [k for k in range(n) if self.held[k]]


def arrival_hooks(model: nn.Module, params: list, bucket_of: list[int]) -> list[tuple[nn.Module, list[int]]]:
    """(module, buckets to wait for before its forward) for the gather / forward overlap: the largest modules
    whose parameters all lie in ONE bucket (a transformer block, the embedding, the final norm, the LM head), plus
    every module that spans buckets and has parameters of its own, for those.  Only modules whose `forward` is
    called can carry a hook: `nn.ModuleList` and `nn.ModuleDict` are containers the parent indexes into, their
    forward never runs, so they are always descended through, also when everything below them shares a bucket
    (all blocks of a small model in one bucket beside the table's)."""
    index = {id(p): bucket_of[i] for i, p in enumerate(params)}
    spans: dict[int, set[int]] = {}
    out: list[tuple[nn.Module, list[int]]] = []

    def span(m: nn.Module) -> set[int]:
        if id(m) not in spans:
            got = {index[id(p)] for p in m.parameters(recurse=False) if id(p) in index}
            for c in m.children():
                got |= span(c)
            spans[id(m)] = got
        return spans[id(m)]

    def walk(m: nn.Module) -> None:
        mine = span(m)
        if not mine:
            return
        container = isinstance(m, (nn.ModuleList, nn.ModuleDict))
        if len(mine) == 1 and not container:
            out.append((m, sorted(mine)))
            return
        own = {index[id(p)] for p in m.parameters(recurse=False) if id(p) in index}
        if own and not container:  # a module that spans buckets and has parameters of its own
            out.append((m, sorted(own)))
        for c in m.children():
            walk(c)

    walk(model)
    return out


class Replica:
    """The training side of one rank: model, flat state, optimizer and the micro-batch graphs."""

    def __init__(self, model: nn.Module, optimizer_factory: Any, group: Any = None, graphs: bool = False,
                 max_graphs: int = 4, compute_dtype: torch.dtype = torch.bfloat16,
                 bucket_bytes: int | None = -1, exchange_alone: bool = False):
        """`model` on the GPU in any floating dtype (its values seed the fp32 master weights);
        `optimizer_factory(module)` builds the optimizer over `module.parameters()`; `compute_dtype` is what
        FSDP2's `param_dtype` is in the reference (bf16; fp32 for exact comparisons).  `bucket_bytes`: size of
        the buckets the gradient exchange is cut into, in bytes of fp32 gradient sums (-1: PRL_BUCKET_MB, 256 MB
        when unset; None: one bucket, exchanged after the pass); with more than one rank and more than one
        bucket the last pass of a step reduce-scatters every bucket while backward is still running."""
        import os

        self.model = model
        from . import fused

        # layers that run back to back in this process, never re-run or sharded one by one: each hands its last
        # residual addition to the next layer's norm kernel (0 when the model's blocks are not the fused ones)
        self.residual_carry = fused.enable_residual_carry(model, os.environ.get("PRL_RESIDUAL_CARRY", "1") != "0")
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        if bucket_bytes == -1:
            bucket_bytes = int(float(os.environ.get("PRL_BUCKET_MB", "256")) * 2**20) or None
        if graphs or not (world > 1 or exchange_alone):
            bucket_bytes = None  # a captured pass is one launch; a single rank exchanges nothing
        self.state = FlatState(model, group, compute_dtype, bucket_bytes, exchange_alone)
        self.device = self.state.flat16.device
        if self.device.type != "cuda":
            raise RuntimeError("Replica: the model must be on a GPU (there is no CPU path)")
        self.optimizer = optimizer_factory(self.state.holder())
        self.graphs = graphs
        self.max_graphs = max_graphs
        self._captured: "OrderedDict[tuple, _Captured]" = OrderedDict()
        self._seen: set[tuple] = set()
        self.capture_errors: dict[tuple, str] = {}
        self.graph_nodes: dict[tuple, tuple[int, int]] = {}  # per captured shape: (nodes, memset nodes)
        self._scale = torch.ones(1, dtype=torch.float32, device=self.device)
        self.replays = 0
        self.eager_runs = 0
        # Without graphs the first pass of a step overwrites the gradient sums instead of adding to them, so nobody
        # clears 4 bytes per parameter between steps or reads them back in that pass (0.9 ms of a 80 ms step).  A
        # captured pass is the same launch whichever pass of a step replays it, so with graphs every pass adds
        # and the optimizer step clears.
        self._starts_sums = False
        # One rank, no graphs: every accumulation launch also leaves the sum of squares of each chunk it has written,
        # and the step's gradient norm is the sum of those few numbers instead of one more pass over 4 bytes per
        # parameter.  `_squares_whole`: a pass that started the sums has written every chunk's entry in this step.
        self._norm_from_chunks = (os.environ.get("PRL_NORM_FROM_CHUNKS", "1") != "0" and not graphs
                                  and not self.state.exchange)
        self._squares_whole = False
        # the chunk table of the accumulation kernel: parameter sizes never change
        lib = C.load()
        chunk = int(lib.prl_accumulate_chunk_elems())
        st = self.state
        tensor_of, start_of = [], []
        for k, n in enumerate(st.numel):
            starts = np.arange(0, n, chunk, dtype=np.int64)
            tensor_of.append(np.full(starts.size, k, dtype=np.int32))
            start_of.append(starts)
        self._chunk_tensor = torch.from_numpy(np.concatenate(tensor_of)).to(self.device)
        self._chunk_start = torch.from_numpy(np.concatenate(start_of)).to(self.device)
        self._numel = torch.tensor(st.numel, dtype=torch.int64).to(self.device)
        self._offsets = torch.tensor(st.offsets, dtype=torch.int64).to(self.device)
        self._chunk_squares = torch.zeros(self._chunk_tensor.numel() if self._norm_from_chunks else 0,
                                          dtype=torch.float32, device=self.device)
        unit_gradient(self.device)  # exists before any capture
        # the exchange overlapped with backward: which chunks belong to which bucket, and the hooks that tell
        # when a bucket's last gradient has landed / when a module is about to read a bucket's weights
        n_chunks = np.cumsum([0] + [len(x) for x in tensor_of])
        for b in st.buckets:
            b.chunk_lo, b.chunk_hi = int(n_chunks[b.first]), int(n_chunks[b.last])
        self.overlap = st.exchange and len(st.buckets) > 1 and not graphs
        self._overlapping = False  # inside the last pass of a step
        self._first_form = False
        self._order = ExchangeOrder([])
        self._src = torch.zeros(len(st.params), dtype=torch.int64, device=self.device)
        self.bucket_launches = 0  # buckets handed to the exchange from inside a backward pass
        if self.overlap:
            table = self.model.get_input_embeddings().weight
            for i, prm in enumerate(st.params):
                if prm is table:
                    st.buckets[st.bucket_of[i]].held = True
                prm.register_post_accumulate_grad_hook(lambda _p, i=i: self._gradient_landed(i))
            self._hook_weight_arrival()

    # ------------------------------------------------------------------ one micro-batch

    def _forward_backward(self, mb: dict[str, Any], plan: Any, loss_config: LossConfig
                          ) -> tuple[Tensor, dict, list, Tensor]:
        """Forward, loss and backward of one row.  Returns the loss, the per-token results, the parameters'
        gradients and the gradient of the looked-up embedding rows [1, T, H].

        The embedding table's own gradient is NOT formed in here.  torch computes it with rocPRIM's
        unique-by-key, which clears two counters with hipMemsetAsync; as a node of a replayed graph that memset
        has left them uncleared (output offsets taken from stale memory: a memory fault).  So the rows are looked
        up without autograd and enter the model as a leaf, and `_embedding_backward` turns their gradient into
        the table's, eagerly, after the pass.  Eager passes do the same, so that both kinds of pass add the same
        numbers in the same order."""
        from .trainer import forward

        with torch.no_grad():
            rows = self.model.get_input_embeddings()(mb["input_ids"])
        rows.requires_grad_(True)
        keep = mb.get("scored_positions")  # the LM head only where the next token is scored (`trainer.scored_rows`)
        logits = forward(self.model, None, mb["position_ids"], plan, rows, positions=keep)
        # normaliser 1 here: the step's own normaliser multiplies the gradients when they are accumulated
        loss, aux = policy_loss(logits, mb["input_ids"], mb["position_ids"], mb["advantages"], mb["logprobs"],
                                mb["loss_mask"], mb["temperature"], loss_config, 1.0, skip_masked=True,
                                positions=keep, max_sample_len=None if plan is None else plan.lmax)
        del logits
        loss.backward(unit_gradient(self.device))
        grads = [p.grad for p in self.state.params]
        return loss.detach(), aux, grads, rows.grad

    # ------------------------------------------------------------------ the exchange inside backward

    def _hook_weight_arrival(self) -> None:
        """A forward pre-hook on every module `arrival_hooks` names: before it runs, the stream waits for the
        all-gathers of the previous step that carry its parameters.  Later buckets are then still arriving while
        earlier ones compute."""
        st = self.state

        def wait_for(which: list[int]) -> Any:
            def hook(_module: nn.Module, _args: Any) -> None:
                for k in which:
                    st.wait_gathered(st.buckets[k])

            return hook

        for module, which in arrival_hooks(self.model, st.params, st.bucket_of):
            module.register_forward_pre_hook(wait_for(which))

    def _gradient_landed(self, i: int) -> None:
        """Post-accumulate hook of parameter `i` (runs on autograd's thread, on the stream of the pass).  Buckets
        go to the exchange in `ExchangeOrder`'s order, so that every rank issues the same collectives in the same
        order whatever order its gradients land in."""
        if not self._overlapping:
            return
        st = self.state
        for k in self._order.landed(st.bucket_of[i]):
            self._launch_bucket(st.buckets[k])
            self.bucket_launches += 1

    def _launch_bucket(self, b: Bucket, reduce: bool = True) -> None:
        """Bucket `b`'s fresh gradients into the flat sums (one launch over the bucket's chunks) and, behind that
        launch, its reduce-scatter; the gradients themselves are dropped."""
        st = self.state
        params = st.params[b.first : b.last]
        ptrs = self._grad_pointers([p.grad for p in params], params)
        self._src[b.first : b.last].copy_(torch.tensor(ptrs, dtype=torch.int64).pin_memory(), non_blocking=True)
        lib = C.load()
        entry = lib.prl_accumulate_first if self._first_form else lib.prl_accumulate
        item32, item64 = self._chunk_tensor.element_size(), self._chunk_start.element_size()
        with C.on_device(self.device):
            C.check(
                entry(C.ptr(self._src), C.dtype_code(st.flat16), C.ptr(self._offsets), C.ptr(self._numel),
                      ctypes_ptr(self._chunk_tensor, b.chunk_lo * item32), ctypes_ptr(self._chunk_start, b.chunk_lo * item64),
                      b.chunk_hi - b.chunk_lo, C.ptr(self._scale), C.ptr(st.acc32), C.stream_ptr(self.device)),
                "prl_accumulate",
            )
        b.launched = True
        for p in params:
            p.grad = None
        if reduce:
            st.reduce_bucket(b)

    def _begin_overlapped_pass(self) -> None:
        self._overlapping = True
        self._first_form = self._starts_sums
        self._starts_sums = False
        st = self.state
        self._order = ExchangeOrder([b.held for b in st.buckets])
        self._order.begin([sum(1 for p in st.params[b.first : b.last] if p.requires_grad) for b in st.buckets])
        for b in st.buckets:
            b.launched = False

    def _end_overlapped_pass(self, input_ids: Tensor, embed_grad: Tensor) -> None:
        """What backward's hooks have not started: buckets with a parameter that got no gradient, and the bucket
        of the embedding table, whose second share (the looked-up rows' gradient) only exists now."""
        self._overlapping = False
        st = self.state
        rest, held = self._order.finish()
        for k in rest:
            self._launch_bucket(st.buckets[k])
        for k in held:
            if not st.buckets[k].launched:
                self._launch_bucket(st.buckets[k], reduce=False)
        self._embedding_backward(input_ids, embed_grad)
        for k in held:
            st.reduce_bucket(st.buckets[k])
        self._first_form = False

    def _exchange_broken(self) -> None:
        """A pass that failed after some of its buckets went to the exchange: the peers are inside collectives
        this rank will never complete.  The job is over either way (the launcher ends every rank once one has
        exited, src/prime_rl/rl.py:641-651; `bench.launch_ranks` and torchrun do the same); aborting the
        communicator, where torch offers it, makes the peers' pending collectives fail at once instead of after
        the launcher's grace period."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return
        try:
            from torch.distributed.distributed_c10d import _abort_process_group

            _abort_process_group()
        except Exception:  # older torch, or a backend without abort (gloo): the launcher's teardown remains
            pass

    def _grad_pointers(self, grads: list, params: list | None = None) -> list[int]:
        dt = self.state.flat16.dtype
        ptrs = []
        for g, p in zip(grads, self.state.params if params is None else params):
            if g is None:
                ptrs.append(0)
                continue
            if g.dtype != dt or not g.is_contiguous() or g.numel() != p.numel():
                raise RuntimeError("Replica: a gradient is not a dense tensor of the compute dtype")
            ptrs.append(g.data_ptr())
        return ptrs

    def _accumulate(self, src: Tensor, first: bool = False) -> None:
        """`first`: the sums start with this pass (nothing cleared them, their old contents are not read)."""
        lib = C.load()
        args = (C.ptr(src), C.dtype_code(self.state.flat16), C.ptr(self._offsets), C.ptr(self._numel),
                C.ptr(self._chunk_tensor), C.ptr(self._chunk_start), self._chunk_tensor.numel(), C.ptr(self._scale),
                C.ptr(self.state.acc32))
        with C.on_device(self.device):
            if self._norm_from_chunks:
                C.check(lib.prl_accumulate_squares(int(first), *args, C.ptr(self._chunk_squares),
                                                   C.stream_ptr(self.device)), "prl_accumulate_squares")
                self._squares_whole = self._squares_whole or first
            else:
                entry = lib.prl_accumulate_first if first else lib.prl_accumulate
                C.check(entry(*args, C.stream_ptr(self.device)), "prl_accumulate")

    def _squared_norm(self) -> Tensor | None:
        """The squared norm of the step's gradient sums from the accumulation kernel's per-chunk sums of squares
        (added in fp64), or None when they do not cover the whole buffer; the step is over either way."""
        whole, self._squares_whole = self._squares_whole, False
        if not whole:
            return None
        return self._chunk_squares.sum(dtype=torch.float64).float()

    def _embedding_backward(self, input_ids: Tensor, grad_rows: Tensor) -> None:
        """The embedding table's share of a micro-batch's gradient, from the gradient of the looked-up rows:
        torch's own (sorted, deterministic) embedding backward, run eagerly, then added to the flat sum like any
        other gradient (for tied weights, on top of the LM head's share that the graph has added)."""
        table = self.model.get_input_embeddings()
        pad = -1 if table.padding_idx is None else int(table.padding_idx)
        grad = torch.ops.aten.embedding_dense_backward(grad_rows.reshape(-1, grad_rows.shape[-1]),
                                                       input_ids.reshape(-1), table.weight.shape[0], pad, False)
        ptrs = [grad.data_ptr() if p is table.weight else 0 for p in self.state.params]
        if not any(ptrs):
            raise RuntimeError("Replica: the embedding table is not among the model's parameters")
        self._accumulate(host_to_device(ptrs, self.device, torch.int64))

    def _drop_grads(self) -> None:
        for p in self.state.params:
            p.grad = None

    @staticmethod
    def _key(mb: dict[str, Any], plan: Any, loss_config: LossConfig) -> tuple:
        sig = None if plan is None else (plan.n, plan.lmax, plan.equal, plan.side_by_side, plan.cu is not None)
        return (tuple(mb["input_ids"].shape), float(mb["temperature"]), sig, loss_config.ratio_type,
                bool(loss_config.ratio_length_norm), float(loss_config.clip_ratio))

    def _capture(self, mb: dict[str, Any], plan: Any, loss_config: LossConfig) -> _Captured:
        cap = _Captured()
        cap.inputs = {k: mb[k].clone() for k in _TENSOR_KEYS}
        cap.inputs["temperature"] = mb["temperature"]
        if plan is not None:  # the tables of the attention layout are inputs too
            import copy

            cap.plan = copy.copy(plan)
            for name in ("gather", "scatter", "cu"):
                if getattr(plan, name) is not None:
                    setattr(cap.plan, name, getattr(plan, name).clone())
        cap.src = torch.zeros(len(self.state.params), dtype=torch.int64, device=self.device)
        cap.graph = torch.cuda.CUDAGraph(keep_graph=True)  # kept, so that its nodes can be looked at before it runs
        with torch.cuda.graph(cap.graph):
            cap.loss, cap.aux, grads, cap.embed_grad = self._forward_backward(cap.inputs, cap.plan, loss_config)
            ptrs = self._grad_pointers(grads)
            self._accumulate(cap.src)
            del grads
            self._drop_grads()
        nodes, memsets = graph_node_counts(cap.graph)
        self.graph_nodes[self._key(mb, plan, loss_config)] = (nodes, memsets)
        if memsets:
            cap.graph.reset()
            raise GraphHoldsMemsets(
                f"Replica: the captured pass holds {memsets} memset nodes among its {nodes} (library calls record "
                "them under capture, e.g. hipBLASLt's stream-K products clearing their flags); such a graph is "
                "never replayed here. Run without graphs (PRL_GRAPHS=0, the default).")
        cap.graph.instantiate()
        # nothing has run yet: the graph reads this table when it is replayed
        cap.src.copy_(torch.tensor(ptrs, dtype=torch.int64))
        return cap

    def micro_batch(self, mb: dict[str, Any], plan: Any, loss_config: LossConfig, last: bool = False
                    ) -> tuple[Tensor, dict[str, Tensor]]:
        """Forward, loss, backward and gradient accumulation of one row that is already on the device.
        Returns the loss (for normaliser 1) and the per-token results, all safe to keep.  `last`: no other pass
        follows in this step, so every bucket's gradient sums are final once its gradients have landed and
        (with several ranks) go to the exchange from inside backward."""
        key = self._key(mb, plan, loss_config)
        cap = self._captured.get(key)
        if cap is None and self.graphs and key in self._seen and key not in self.capture_errors:
            try:
                cap = self._capture(mb, plan, loss_config)
            except GraphHoldsMemsets:
                self._drop_grads()
                raise
            except RuntimeError as e:  # something in this shape's forward cannot be captured: it stays eager
                self.capture_errors[key] = repr(e)
                self._drop_grads()
                cap = None
            else:
                self._captured[key] = cap
                while len(self._captured) > self.max_graphs:
                    self._captured.popitem(last=False)
        if len(self._seen) > 4096:  # rows of ever-changing lengths: forget, rather than remember every shape
            self._seen.clear()
        self._seen.add(key)
        if cap is not None:
            self._captured.move_to_end(key)
            for k in _TENSOR_KEYS:
                cap.inputs[k].copy_(mb[k])
            if cap.plan is not None:
                for name in ("gather", "scatter", "cu"):
                    if getattr(plan, name) is not None:
                        getattr(cap.plan, name).copy_(getattr(plan, name))
            cap.graph.replay()
            self._embedding_backward(cap.inputs["input_ids"], cap.embed_grad)
            self.replays += 1
            return cap.loss.clone(), {k: cap.aux[k].clone() for k in _AUX_KEYS}
        if self.overlap and last:
            self._begin_overlapped_pass()
            try:
                loss, aux, grads, embed_grad = self._forward_backward(mb, plan, loss_config)
            except BaseException:
                self._overlapping = False
                if any(b.launched for b in self.state.buckets):
                    self._exchange_broken()
                raise
            self._overlapping = False
            del grads
            self._end_overlapped_pass(mb["input_ids"], embed_grad)
            del embed_grad
            self._drop_grads()
            self.eager_runs += 1
            return loss, aux
        loss, aux, grads, embed_grad = self._forward_backward(mb, plan, loss_config)
        src = host_to_device(self._grad_pointers(grads), self.device, torch.int64)
        self._accumulate(src, first=self._starts_sums)
        self._starts_sums = False
        self._embedding_backward(mb["input_ids"], embed_grad)
        del grads, embed_grad
        self._drop_grads()
        self.eager_runs += 1
        return loss, aux

    def wait_parameters(self) -> None:
        """The current stream waits for every all-gather still in flight: call before reading the model's
        parameters outside a forward pass (publishing weights, snapshots)."""
        self.state.wait_gathered()

    # ------------------------------------------------------------------ one optimizer step

    def prepare_step(self, micro_batches: list[MicroBatch], loss_config: LossConfig,
                     recomputed_errors: list[Tensor] | None = None, join_tokens: int | None = None,
                     skip_unscored: bool = True) -> dict[str, Any]:
        """The host's share of a step before anything runs: fillers left out (`trainer.scored`), samples without a
        gradient sorted out (`trainer.split_by_gradient`), rows joined (`trainer.join_micro_batches`).  Touches
        host tensors only (unless `recomputed_errors`, device tensors, are given), so the caller may do it for
        the NEXT step while this step's kernels still run; hand the result to `train_step(prepared=...)`."""
        from .trainer import (MERGE_MAX_TOKENS, join_micro_batches, join_per_row, local_loss_scale, scored,
                              split_by_gradient)

        join_tokens = MERGE_MAX_TOKENS if join_tokens is None else join_tokens
        n_given = len(micro_batches)
        inv_scale = 1.0 / max(float(local_loss_scale(micro_batches, loss_config)), 1.0)
        token_mode = loss_config.ratio_type == "token"
        given_row = list(range(n_given))  # the given row every micro-batch still in the list came from
        if skip_unscored and token_mode:
            keep = scored(micro_batches)
            micro_batches = [micro_batches[i] for i in keep]
            given_row = keep
            if recomputed_errors is not None:
                recomputed_errors = [recomputed_errors[i] for i in keep]
        # samples nothing can be learned from (advantage 0 wherever they are scored) take a forward pass only
        forward_only: list[MicroBatch] = []
        forward_row: list[int] = []
        if skip_unscored and token_mode and recomputed_errors is None and not self.graphs:
            split = split_by_gradient(micro_batches)
            if split is not None:
                micro_batches, rows_a, forward_only, rows_b = split
                forward_row = [given_row[i] for i in rows_b]
                given_row = [given_row[i] for i in rows_a]

        def units_of(mbs: list[MicroBatch]) -> tuple[list[Any] | None, list[MicroBatch]]:
            rows = join_micro_batches(mbs, join_tokens) if token_mode and join_tokens > 0 else None
            return rows, ([r.micro_batch for r in rows] if rows is not None else mbs)

        rows_b, units_b = units_of(forward_only)
        rows, units = units_of(micro_batches)
        if rows is not None and recomputed_errors is not None:
            recomputed_errors = join_per_row(recomputed_errors, rows)
        return {"n_given": n_given, "inv_scale": inv_scale, "given_row": given_row, "forward_row": forward_row,
                "rows_b": rows_b, "units_b": units_b, "rows": rows, "units": units,
                "recomputed_errors": recomputed_errors, "ratio_type": loss_config.ratio_type}

    def train_step(self, micro_batches: list[MicroBatch], loss_config: LossConfig, max_norm: float | None = 1.0,
                   scheduler: Any = None, samples: int | None = None,
                   recomputed_errors: list[Tensor] | None = None, join_tokens: int | None = None,
                   skip_unscored: bool = True, prepared: dict[str, Any] | None = None,
                   before_statistics: Any = None) -> Any:
        """One optimizer step over this rank's micro-batches (src/prime_rl/trainer/rl/train.py:259-364);
        same results and the same `StepResult` as `trainer.train_step`, micro-batches joined the same way and
        fillers left out the same way (`trainer.scored`).  `prepared`: what `prepare_step` returned for these
        micro-batches, if the caller has it already.  `before_statistics()` is called when the whole step has been
        handed to the GPU, just before the host waits for the step's statistics: the place for host work that
        can run beside the GPU (the trainer loads and prepares the next step's rollouts there)."""
        from .trainer import (StepResult, attention_plan, add_ratio_statistics, forward, losses_per_given_row,
                              ones_as_recomputed_error, restore_row_starts, scored_rows)

        t0 = time.perf_counter()
        if prepared is None:
            prepared = self.prepare_step(micro_batches, loss_config, recomputed_errors, join_tokens, skip_unscored)
        elif prepared["ratio_type"] != loss_config.ratio_type or prepared["n_given"] != len(micro_batches):
            raise ValueError("train_step: `prepared` belongs to other micro-batches or another loss configuration")
        n_given, inv_scale = prepared["n_given"], prepared["inv_scale"]
        given_row, forward_row = prepared["given_row"], prepared["forward_row"]
        rows_b, units_b, rows, units = prepared["rows_b"], prepared["units_b"], prepared["rows"], prepared["units"]
        recomputed_errors = prepared["recomputed_errors"]
        self._starts_sums = not self.graphs
        self._squares_whole = False
        self._scale.fill_(inv_scale)
        tensors = Tensors()
        vocab = int(self.model.config.vocab_size)
        loss_parts: list[Tensor] = []
        loss_rows: list[int] = []
        state = {"tokens": 0, "passes": 0}

        def book(joined: Any, mb: dict[str, Any], loss: Tensor, aux: dict[str, Tensor], came_from: list[int],
                 error: Tensor | None) -> None:
            """Statistics and per-row losses of one pass (many small launches: after the pass's own work)."""
            if joined is not None:
                losses = restore_row_starts(joined, mb, aux, loss_config, vocab, inv_scale)
                mask = host_to_device(joined.loss_mask, self.device)
            else:
                losses, mask = (loss * inv_scale).reshape(1), mb["loss_mask"]
            tensors.add("probs", aux["logprobs"], mask, exp=True)
            tensors.add("old_probs", mb["logprobs"], mask, exp=True)
            tensors.add("entropy", aux["entropy"], mask)
            loss_parts.append(losses)
            loss_rows.extend(came_from)
            if error is not None:
                tensors.add("recomputed_logprob_error", error, mask)
            add_ratio_statistics(tensors, aux, mask, loss_config)
            state["tokens"] += int(mb["input_ids"].numel()) - (joined.padding if joined is not None else 0)
            state["passes"] += 1

        def pieces_of(rows: list[Any] | None, k: int, came_from: list[int]) -> tuple[list[int], int]:
            n = len(rows[k].lengths) if rows is not None else 1
            at = sum(len(r.lengths) for r in rows[:k]) if rows is not None else k
            return came_from[at : at + n], n

        # forward only first, so that the last pass of the step is the one the exchange hangs on
        for k, host_mb in enumerate(units_b):
            mb = to_device(host_mb, self.device)
            plan = attention_plan(host_mb, self.device)
            keep = scored_rows(host_mb, loss_config, self.model, self.device)
            with torch.no_grad():
                logits = forward(self.model, mb["input_ids"], mb["position_ids"], plan, positions=keep)
                loss, aux = policy_loss(logits, mb["input_ids"], mb["position_ids"], mb["advantages"], mb["logprobs"],
                                        mb["loss_mask"], mb["temperature"], loss_config, 1.0, skip_masked=True,
                                        positions=keep, max_sample_len=None if plan is None else plan.lmax)
                del logits
            book(rows_b[k] if rows_b is not None else None, mb, loss, aux, pieces_of(rows_b, k, forward_row)[0], None)

        grad_norm = None
        for step_idx, host_mb in enumerate(units):
            mb = to_device(host_mb, self.device)
            if not self.graphs:  # (a captured pass is keyed by its shapes; the number of scored rows changes)
                keep = scored_rows(host_mb, loss_config, self.model, self.device)
                if keep is not None:
                    mb["scored_positions"] = keep
            loss, aux = self.micro_batch(mb, attention_plan(host_mb, self.device), loss_config,
                                         last=step_idx == len(units) - 1)
            if step_idx == len(units) - 1:
                # The end of the step goes to the GPU before the last pass's bookkeeping below, which is many
                # small launches (and one host synchronisation): 4 ms of norm and optimizer kernels then run
                # while the host issues them, instead of the GPU waiting 0.3-0.5 ms for each in turn.
                grad_norm = self.state.finish_step(self.optimizer, max_norm, keep_sums=not self.graphs,
                                                   gathers_in_flight=self.overlap, squared_norm=self._squared_norm())
            book(rows[step_idx] if rows is not None else None, mb, loss, aux, pieces_of(rows, step_idx, given_row)[0],
                 recomputed_errors[step_idx] if recomputed_errors is not None else None)
        # one loss per given micro-batch, as the reference logs it; 0 for those not run (nothing scored)
        tensors.add("loss", losses_per_given_row(loss_parts, loss_rows, n_given, self.device))
        if not units:  # nothing to learn from on this rank: its share of the mean is zero, not last step's sums
            self.state.acc32.zero_()
            self._squares_whole = False
            if self.overlap:
                # the other ranks start their reduce-scatters from inside a backward pass: every bucket but the
                # embedding table's back to front, then that one; collectives have to come in one order everywhere
                for k in ExchangeOrder([b.held for b in self.state.buckets]).sequence():
                    self.state.reduce_bucket(self.state.buckets[k])
            grad_norm = self.state.finish_step(self.optimizer, max_norm, keep_sums=not self.graphs,
                                               gathers_in_flight=self.overlap)
        if scheduler is not None:
            scheduler.step()
        if before_statistics is not None:
            before_statistics()
        stats = tensors.compute_stats()
        if recomputed_errors is None:
            ones_as_recomputed_error(stats)
        seconds = time.perf_counter() - t0
        if samples is None:
            samples = n_given
        return StepResult(loss=stats.get("loss/mean", float("nan")), grad_norm=float(grad_norm), stats=stats,
                          seconds=seconds, tokens=state["tokens"], samples=samples, passes=state["passes"])


def whole_model_bytes(model: nn.Module, tokens_per_pass: int, world: int = 1) -> int:
    """What one rank of `Replica` holds at the peak of a step WITHOUT activation checkpointing, in bytes, for a
    forward / backward pass over `tokens_per_pass` tokens: the bf16 model and the fp32 gradient sums whole, fp32
    master weights and the two AdamW moments as a 1 / world shard, every block's saved activations, and the logits
    with their gradient.  The activations of a block are counted as the bf16 tensors its autograd functions keep:
    6 hidden-wide ones (inputs of the two norms, their outputs, the attention output, the residual), the q / k / v
    product and the rotated queries and keys, the gate / up product and the activation's output.  Checked against
    the allocator's peak for Qwen2.5-7B at 16384 tokens on one rank: 205 GiB measured, 215 GiB by this count."""
    cfg = model.config
    n = sum(p.numel() for p in model.parameters())
    h, inter = int(cfg.hidden_size), int(cfg.intermediate_size)
    head = int(getattr(cfg, "head_dim", None) or h // cfg.num_attention_heads)
    q_wide, kv_wide = head * int(cfg.num_attention_heads), head * int(cfg.num_key_value_heads)
    per_token_and_block = 2 * (6 * h + 2 * (q_wide + 2 * kv_wide) + q_wide + 3 * inter)
    state = 2 * n + 4 * n + (12 * n + world - 1) // world
    activations = tokens_per_pass * int(cfg.num_hidden_layers) * per_token_and_block
    logits = tokens_per_pass * int(cfg.vocab_size) * (2 + 2) + tokens_per_pass * h * 8
    return state + activations + logits


HBM_SHARE = 0.85  # of a GPU's memory that a whole-model step may be estimated at (the rest: workspaces, fragmentation)


def eligible(model: nn.Module, optim_type: str, activation_checkpointing: bool = False,
             tokens_per_pass: int = 16384, world: int = 1, hbm_bytes: int | None = None,
             extra_bytes: int = 0) -> str | None:
    """None when `Replica` can train this model, otherwise the reason it cannot (then FSDP2 does).

    Activation checkpointing, as the reference's recipes for 80 GB GPUs ask for it (`model.ac`), trades a second
    forward pass for memory and changes no result.  `Replica` does not re-run blocks inside backward, so a model
    that asks for it is taken only if a step WITHOUT checkpointing fits the GPU (`whole_model_bytes` against
    `HBM_SHARE` of `hbm_bytes`, default: the 288 GB of an MI355X): then the recipe's request is simply not needed.
    `extra_bytes`: what else the caller keeps on the GPU (a second copy of the model for recomputed log-probabilities)."""
    from .trainer import keeps_parameters_gathered

    if not keeps_parameters_gathered(model):
        return "the bf16 parameters do not fit the 32 GB a rank keeps whole"
    if optim_type not in ("adamw", "sgd"):
        return f"optimizer {optim_type!r} is not element-wise over a flat shard"
    if activation_checkpointing:
        hbm = 288 * 10**9 if hbm_bytes is None else int(hbm_bytes)
        need = whole_model_bytes(model, tokens_per_pass, world) + int(extra_bytes)
        if need > HBM_SHARE * hbm:
            return (f"activation checkpointing is needed (a step without it is estimated at {need / 2**30:.0f} GiB of "
                    f"{hbm / 2**30:.0f}), and it re-runs blocks inside backward")
    return None
