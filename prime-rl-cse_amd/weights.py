"""Trainer -> inference weight publication on a side HIP stream
(counterpart of src/prime_rl/trainer/weights.py:96-255).

The reference gathers every parameter on the compute stream, copies it to the
host with a blocking copy, runs a barrier, and only then lets the step go on
(`weights.py:130-143`, called at `trainer/rl/train.py:163-166`).  Here:

  1. SNAPSHOT  on a side stream, ordered after the optimizer step by an event:
               every local shard is cast to bf16 into memory of its own.  The
               compute stream only ever waits for this (a read of 1/W of the
               model), and only if it reaches the next optimizer step first.
  2. GATHER    on the side stream: the bf16 shards are all-gathered over RCCL
               (after the downcast, as the reference does, so half the bytes
               travel), overlapping the next step's forward and backward.
  3. COPY-OUT  rank 0 copies the gathered tensors into pinned host memory with
               asynchronous copies on the side stream and records an event.
               (Tensors that are whole on this GPU skip 2 and share ONE staging
               buffer on the device and one in pinned memory: one multi-tensor
               copy, one transfer.)
  4. WRITE     a host thread waits for that event, writes `pytorch_model.tmp`
               with torch.save and renames it to `pytorch_model.bin`
               (`weights.py:169-174`): the file layout, the atomic rename and the
               directory scheme `weights/step_N/` are the reference's, so its
               orchestrator and inference worker read the result unchanged.

Consumers on the same node that can take device memory register a callback
with `add_device_subscriber`; it is called on the side stream with the gathered
bf16 tensors before the copy-out, which is where a device-to-device hand-off to
an inference engine plugs in.
"""

from __future__ import annotations

import contextlib
import shutil
import threading
from pathlib import Path
from typing import Any, Callable

import torch
import torch.distributed as dist
from torch import Tensor, nn

try:  # DTensor is present in every torch this package supports; keep CPU-only tools importable regardless
    from torch.distributed.tensor import DTensor
except Exception:  # pragma: no cover
    DTensor = ()  # type: ignore[assignment]

from .world import get_world


def get_weights_dir(output_dir: Path) -> Path:
    return Path(output_dir) / "weights"


def get_step_path(path: Path, step: int) -> Path:
    return Path(path) / f"step_{step}"


def get_weight_ckpt_model_path(weights_dir: Path, step: int) -> Path:
    return Path(weights_dir) / f"step_{step}" / "pytorch_model.bin"


def wait_for_weight_checkpoint(weights_dir: Path, step: int, interval: float = 1.0, timeout: float | None = None):
    """What the orchestrator does before it tells the inference server to reload
    (src/prime_rl/orchestrator/utils.py:53-55)."""
    from .data import wait_for_path

    wait_for_path(get_weight_ckpt_model_path(weights_dir, step), interval, timeout)


class PendingPublication:
    """Handle of one publication in flight."""

    def __init__(self, step: int, path: Path):
        self.step, self.path = step, path
        self.snapshot_done: Any = None  # event: local shards copied, parameters may change again
        self._thread: threading.Thread | None = None
        self._slot: Tensor | None = None  # the pinned staging buffer this publication writes from
        self.error: BaseException | None = None

    def wait(self, timeout: float | None = None) -> Path:
        """Block until the file is in place (rank 0) or the gather has been issued (other ranks)."""
        if self._thread is not None:
            self._thread.join(timeout)
            if self._thread.is_alive():
                raise TimeoutError(f"weight publication of step {self.step} still running")
        if self.error is not None:
            raise self.error
        return self.path


def _flat_staging_enabled() -> bool:
    """PRL_PUBLISH_FLAT=0 keeps the per-tensor copies for whole tensors too (for A/B measurements)."""
    import os

    return os.environ.get("PRL_PUBLISH_FLAT", "1") != "0"


class WeightPublisher:
    """Publishes HF-compatible bf16 weight files without stalling the step."""

    def __init__(self, output_dir: Path, interval: int | None = None, ckpt_interval: int | None = None,
                 async_level: int = 2, device: torch.device | str | None = None, max_in_flight: int = 3,
                 crc32: bool = True):
        """`max_in_flight` publications may be on their way to the disk at once; a further `publish` first waits
        for the oldest (steps faster than the disk would otherwise pile up pinned host copies of the weights
        without bound, and pinning fresh host memory for every publication costs more than the copy itself:
        ~100 ms per 3 GB).  Three writers keep up with a publication every ~90 ms for a 1.2 GB model (two: 135 ms;
        more than three gain nothing, `tools/publish_profile.py`).

        `crc32=False` leaves the zip records' CRC32 out of the files: torch.save of a 1.2 GB model takes 267 ms
        with it and 118 ms without, and torch.load (what the inference worker and transformers read these files
        with) never looks at it; other zip tools may complain about its absence.  torch has ONE switch for the
        whole process, so this publisher turns it off for as long as it lives and `close()` puts back what it
        found; the default leaves torch's setting alone."""
        self.max_in_flight = max(1, int(max_in_flight))
        self._crc32_before: bool | None = None
        if not crc32 and hasattr(torch.serialization, "set_crc32_options"):
            self._crc32_before = bool(torch.serialization.get_crc32_options())
            torch.serialization.set_crc32_options(False)
        self.weights_dir = get_weights_dir(output_dir)
        self.interval, self.ckpt_interval, self.async_level = interval, ckpt_interval, async_level
        self.device = torch.device(device) if device is not None else None
        self._is_master = get_world().is_master
        self._stream: Any = None
        self._subscribers: list[Callable[[dict[str, Tensor], int], None]] = []
        self._pending: list[PendingPublication] = []
        self._slots: list[Tensor] = []  # pinned staging buffers of the whole-tensor path
        self._cleaners: list[threading.Thread] = []

    # ------------------------------------------------------------------ plumbing

    def add_device_subscriber(self, fn: Callable[[dict[str, Tensor], int], None]) -> None:
        """`fn(state, step)` runs on the publication stream with the gathered bf16 tensors still on the device."""
        self._subscribers.append(fn)

    def _side_stream(self):
        if self.device is None or self.device.type != "cuda":
            return None
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)  # default priority on purpose
        return self._stream

    @staticmethod
    def _layout(state: dict[str, Tensor], dtype: torch.dtype) -> tuple[list[tuple], int]:
        """(name, shape, offset, elements) of every tensor in one staging buffer of `dtype`, slices 256-byte
        aligned; tensors that share their memory (tied embeddings) share their slice, so the file keeps them tied
        as `torch.save` of the model's own state dict would.  One element type per buffer: `torch.save` refuses
        tensors that view one storage as different types."""
        unit = max(1, 256 // torch.empty((), dtype=dtype).element_size())
        layout, off, seen = [], 0, {}
        for name, v in state.items():
            n = v.numel()
            key = (v.data_ptr(), tuple(v.shape), tuple(v.stride()), v.dtype)
            if key in seen and n:
                layout.append((name, tuple(v.shape), seen[key], n))
                continue
            seen[key] = off
            layout.append((name, tuple(v.shape), off, n))
            off += (n + unit - 1) // unit * unit
        return layout, max(off, 1)

    @staticmethod
    def _views(buffer: Tensor, layout: list[tuple]) -> dict[str, Tensor]:
        return {name: buffer[off : off + n].view(shape) for name, shape, off, n in layout}

    _pinned_refused = False  # set once pinned host memory has run out on this machine

    @classmethod
    def _host_buffer(cls, shape: Any, dtype: torch.dtype, zero: bool = False) -> Tensor:
        """Host memory for a copy-out: pinned (the copy then runs beside the compute stream), or, once the machine
        refuses to pin more (its locked-memory limit, not this process's doing), ordinary memory: the copy then
        holds the side stream up, the publication still happens."""
        make = torch.zeros if zero else torch.empty
        if not cls._pinned_refused:
            try:
                return make(shape, dtype=dtype, pin_memory=True)
            except RuntimeError as exc:  # "HIP error: out of memory" from the pinning call
                cls._pinned_refused = True
                import logging

                logging.getLogger(__name__).warning("weight publication: no more pinned host memory (%s); "
                                                    "copying out through pageable memory from here on", exc)
        return make(shape, dtype=dtype)

    def _host_slot(self, total: int, dtype: torch.dtype) -> Tensor:
        """A pinned buffer of `total` elements that no publication in flight is using (at most `max_in_flight`
        are, and `publish` has waited for the oldest): buffers are kept and reused, pinning memory costs more
        than the copy."""
        busy = {id(p._slot) for p in self._pending if p._slot is not None and p._thread is not None
                and p._thread.is_alive()}
        fits = lambda b: b.numel() == total and b.dtype == dtype
        for buf in self._slots:
            if id(buf) not in busy and fits(buf):
                return buf
        buf = self._host_buffer(total, dtype, zero=True)  # zeros: the padding between slices is written too
        self._slots = [b for b in self._slots if id(b) in busy or fits(b)] + [buf]
        return buf

    # ------------------------------------------------------------------ publication

    def publish(self, model: nn.Module, step: int, dtype: torch.dtype = torch.bfloat16, tokenizer: Any = None,
                state_dict: dict[str, Tensor] | None = None) -> PendingPublication:
        """Start publishing the current weights as `weights/step_<step>/pytorch_model.bin`.

        Returns at once.  Call `guard_parameters()` (or `pending.snapshot_done.synchronize()`) before the next
        in-place update of the parameters; the trainer loop in this package does."""
        alive = [p for p in self._pending if p._thread is not None and p._thread.is_alive()]
        while len(alive) >= self.max_in_flight:
            alive.pop(0)._thread.join()
        pending = PendingPublication(step, get_weight_ckpt_model_path(self.weights_dir, step))
        state = state_dict if state_dict is not None else model.state_dict()
        side = self._side_stream()
        ctx = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
        if side is not None:
            side.wait_stream(torch.cuda.current_stream(self.device))  # after the optimizer step
        whole = side is not None and all(v.is_cuda and not isinstance(v, DTensor) for v in state.values()) \
            and any(v.is_floating_point() for v in state.values()) and _flat_staging_enabled()
        with ctx, torch.no_grad():
            host: dict[str, Tensor] = {}
            copied = None
            if whole:
                # Whole tensors on this GPU (the whole-model trainer, or one rank): ONE staging buffer on the
                # device and ONE in pinned host memory, every tensor a 256-byte-aligned slice of them.  The
                # snapshot is a multi-tensor copy, the copy-out a single transfer, and the host does a handful
                # of calls instead of three per tensor (21 ms per publication for the 311 tensors of a 0.6B model).
                floating = {k: v for k, v in state.items() if v.is_floating_point()}
                others = {k: v.detach().clone() for k, v in state.items() if not v.is_floating_point()}
                layout, total = self._layout(floating, dtype)
                flat = torch.empty(total, dtype=dtype, device=self.device)
                staged = self._views(flat, layout)
                first = {}  # one copy per slice: tied tensors share theirs
                for name, _, off, n in layout:
                    if n:  # (an empty tensor takes no room: the next one starts at the same offset)
                        first.setdefault(off, name)
                torch._foreach_copy_([staged[k] for k in first.values()], [floating[k].detach() for k in first.values()])
                full = {k: (staged[k] if k in staged else others[k]) for k in state}  # the state dict's own order
                pending.snapshot_done = torch.cuda.Event()
                pending.snapshot_done.record(side)
                for fn in self._subscribers:
                    fn(full, step)
                if self._is_master:
                    slot = self._host_slot(total, dtype)
                    slot.copy_(flat, non_blocking=True)
                    on_host = self._views(slot, layout)
                    for k, v in others.items():  # the odd integer buffer: a tensor of its own
                        buf = self._host_buffer(v.shape, v.dtype)
                        buf.copy_(v, non_blocking=True)
                        on_host[k] = buf
                    host = {k: on_host[k] for k in state}
                    copied = torch.cuda.Event()
                    copied.record(side)
                    pending._slot = slot
            else:
                # 1. snapshot: bf16 copies of the local shards
                # (always a copy: parameters that already are in `dtype` would otherwise be read by the copy-out
                # while the next optimizer step rewrites them)
                snap = {k: (v.detach().to(dtype, copy=True) if v.is_floating_point() else v.detach().clone())
                        for k, v in state.items()}
                if side is not None:
                    pending.snapshot_done = torch.cuda.Event()
                    pending.snapshot_done.record(side)
                # 2. gather (after the downcast: half the bytes on the wire)
                full = {}
                for k, v in snap.items():
                    full[k] = v.full_tensor() if isinstance(v, DTensor) else v
                for fn in self._subscribers:
                    fn(full, step)
                # 3. copy-out on rank 0: the floating tensors into slices of ONE pinned buffer that is kept and
                # reused (a pinned allocation per tensor and publication is rounded up to a power of two by the
                # host allocator and ran a 7B model out of lockable memory on the third publication)
                if self._is_master:
                    together = {k: v for k, v in full.items()
                                if v.device.type == "cuda" and v.is_floating_point() and v.dtype == dtype}
                    on_host: dict[str, Tensor] = {}
                    if together:
                        layout, total = self._layout(together, dtype)
                        slot = self._host_slot(total, dtype)
                        on_host = self._views(slot, layout)
                        pending._slot = slot
                    for k, v in full.items():
                        if k in on_host:
                            on_host[k].copy_(v, non_blocking=True)
                            host[k] = on_host[k]
                        elif v.device.type == "cuda":
                            buf = self._host_buffer(v.shape, v.dtype)
                            buf.copy_(v, non_blocking=True)
                            host[k] = buf
                        else:
                            host[k] = v
                    if side is not None:
                        copied = torch.cuda.Event()
                        copied.record(side)
        # 4. write on a host thread
        if self._is_master:
            config = getattr(model, "config", None) if model is not None else None
            gen_config = getattr(model, "generation_config", None) if model is not None else None

            def write() -> None:
                try:
                    if copied is not None:
                        copied.synchronize()
                    step_path = get_step_path(self.weights_dir, step)
                    step_path.mkdir(parents=True, exist_ok=True)
                    tmp = pending.path.with_suffix(".tmp")
                    torch.save(host, tmp)
                    tmp.rename(pending.path)  # existence of the .bin means the file is complete
                    if config is not None and hasattr(config, "save_pretrained"):
                        config.save_pretrained(step_path)
                    if gen_config is not None and hasattr(gen_config, "save_pretrained"):
                        gen_config.save_pretrained(step_path)
                    if tokenizer is not None:
                        tokenizer.save_pretrained(step_path)
                except BaseException as exc:  # surfaced by wait()
                    pending.error = exc

            pending._thread = threading.Thread(target=write, name=f"weight-publication-{step}")
            pending._thread.start()
        self._pending = [p for p in self._pending if p._thread is not None and p._thread.is_alive()] + [pending]
        return pending

    def guard_parameters(self) -> None:
        """Make the compute stream wait until every snapshot in flight has read the parameters.
        Call before an in-place parameter update (the optimizer step)."""
        if self.device is None or self.device.type != "cuda":
            return
        for p in self._pending:
            if p.snapshot_done is not None:
                torch.cuda.current_stream(self.device).wait_event(p.snapshot_done)

    def wait_all(self) -> None:
        for p in list(self._pending):
            p.wait()
        for t in self._cleaners:
            t.join()
        self._cleaners = []

    def close(self) -> None:
        """Every publication written, every clean-up done, and torch's CRC32 switch as this publisher found it."""
        self.wait_all()
        if self._crc32_before is not None:
            torch.serialization.set_crc32_options(self._crc32_before)
            self._crc32_before = None

    # ------------------------------------------------------------------ clean-up

    def keep_step(self, step: int) -> bool:
        """Whether `weights/step_<step>` must stay: evaluation steps (`step % interval == 0`) and the
        `async_level + 1` steps leading up to a full checkpoint, which the logprob models of a resumed run
        need (src/prime_rl/trainer/weights.py:215-223)."""
        keep_for_eval = bool(self.interval) and step % self.interval == 0
        keep_for_ckpt = bool(self.ckpt_interval) and (
            (self.ckpt_interval - (step % self.ckpt_interval)) % self.ckpt_interval <= self.async_level
        )
        return keep_for_eval or keep_for_ckpt

    def maybe_clean(self, step: int, wait: bool = True) -> Path | None:
        """Consider deleting the directory of `async_level + 1` steps ago
        (src/prime_rl/trainer/weights.py:211-228).  Returns the path removed, if any.  `wait=False` removes it on
        a thread of its own, as the reference does (`weights.py:236-242`): unlinking a 1.2 GB file takes the
        caller 23 ms; `wait_all` waits for those threads too."""
        old = max(step - (self.async_level + 1), 0)
        if self.keep_step(old):
            return None
        path = get_step_path(self.weights_dir, old)
        writers = [p._thread for p in self._pending if p.step == old and p._thread is not None]

        def remove() -> None:
            for t in writers:  # a publication of that step still being written: let it finish, then remove it
                t.join()
            shutil.rmtree(path, ignore_errors=True)

        if wait:
            remove()
        else:
            self._cleaners = [t for t in self._cleaners if t.is_alive()]
            t = threading.Thread(target=remove, name=f"weight-cleanup-{old}")
            t.start()
            self._cleaners.append(t)
        return path


# --------------------------------------------------------------------------- device-to-device hand-off


def _aligned(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


class WeightBroadcast:
    """Trainer -> inference weights without the file system: ONE broadcast of a flat
    bucket per publication, over RCCL (xGMI inside a node) on the publication stream.

    The reference moves weights through `weights/step_N/pytorch_model.bin` and an HTTP
    call that makes every inference worker `torch.load` the file
    (src/prime_rl/trainer/weights.py:130-209, src/prime_rl/orchestrator/client.py:63-86,
    src/prime_rl/inference/vllm/worker.py:14-30).  This is the device path beside it:
    the trainer's rank 0 and the inference ranks form one process group; the sender
    packs the gathered bf16 tensors into one contiguous bucket (256-byte aligned
    slots) and broadcasts the bucket, preceded by a 2-word header (step, bytes).
    xGMI links are point to point, so one large message per publication is the shape
    that uses them well: 1.2 GB for Qwen3-0.6B instead of 311 small ones.

    The receiver owns TWO buckets and fills them alternately: the tensors `receive()`
    returned last time stay valid (the engine can keep decoding with them) while the
    next publication lands in the other bucket.

    Sender:    ch = WeightBroadcast(group, src=0, device=dev); publisher.add_device_subscriber(ch.send)
    Receiver:  ch = WeightBroadcast(group, src=0, device=dev); step, state = ch.receive()
    The first exchange also carries the manifest (names, shapes, dtypes, offsets) as a
    pickled object through the group's host path; later ones are header + bucket only.
    A header with step -1 (`close()`) tells receivers to stop.
    """

    def __init__(self, group: Any, src: int, device: torch.device | str):
        self.group, self.src, self.device = group, src, torch.device(device)
        if self.device.type == "cuda" and dist.get_backend(group) == "gloo":
            raise ValueError("WeightBroadcast: a gloo group moves host tensors only (device='cpu'); "
                             "device buckets need an RCCL (\"nccl\") group")
        self.is_sender = dist.get_rank() == src
        self.is_member = dist.get_rank(group) >= 0 if group is not None else True
        self.manifest: list[tuple[str, tuple[int, ...], torch.dtype, int, int]] | None = None
        self.total_bytes = 0
        self._buckets: list[Tensor] = []
        self._turn = 0

    # -- both sides

    def _set_manifest(self, manifest: list[tuple[str, tuple[int, ...], torch.dtype, int, int]]) -> None:
        self.manifest = manifest
        self.total_bytes = max((off + _aligned(nbytes) for _, _, _, off, nbytes in manifest), default=0)
        n = 1 if self.is_sender else 2
        self._buckets = [torch.empty(max(self.total_bytes, 1), dtype=torch.uint8, device=self.device)
                         for _ in range(n)]

    def _views(self, bucket: Tensor) -> dict[str, Tensor]:
        out = {}
        for name, shape, dtype, off, nbytes in self.manifest or []:
            out[name] = bucket[off : off + nbytes].view(dtype).view(shape)
        return out

    # -- sender

    def send(self, state: dict[str, Tensor], step: int) -> None:
        """Broadcast `state` (tensors on `device`).  Runs on the caller's current stream; as a
        `WeightPublisher` device subscriber that is the publication side stream.  Ranks outside the
        group return at once."""
        if not self.is_member:
            return
        if not self.is_sender:
            raise RuntimeError("WeightBroadcast.send called on a rank that is not the source")
        if self.manifest is None:
            manifest, off = [], 0
            for name, t in state.items():
                nbytes = t.numel() * t.element_size()
                manifest.append((name, tuple(t.shape), t.dtype, off, nbytes))
                off += _aligned(nbytes)
            dist.broadcast_object_list([manifest], src=self.src, group=self.group)
            self._set_manifest(manifest)
        elif [m[0] for m in self.manifest] != list(state.keys()):
            raise RuntimeError("WeightBroadcast: the set of tensors changed between publications")
        bucket = self._buckets[0]
        views = self._views(bucket)
        with torch.no_grad():
            for name, t in state.items():
                views[name].copy_(t, non_blocking=True)
        header = torch.tensor([step, self.total_bytes], dtype=torch.int64).to(self.device, non_blocking=True)
        dist.broadcast(header, src=self.src, group=self.group)
        dist.broadcast(bucket, src=self.src, group=self.group)

    def close(self) -> None:
        if self.is_member and self.is_sender:
            header = torch.tensor([-1, 0], dtype=torch.int64).to(self.device)
            if self.manifest is None:
                dist.broadcast_object_list([[]], src=self.src, group=self.group)
                self._set_manifest([])
            dist.broadcast(header, src=self.src, group=self.group)

    # -- receiver

    def receive(self) -> tuple[int, dict[str, Tensor]] | None:
        """Block until the next publication has been enqueued into this rank's idle bucket; returns
        (step, tensors viewing that bucket), or None after `close()`.  With RCCL the tensors are ready
        in stream order on the current stream."""
        if self.is_sender:
            raise RuntimeError("WeightBroadcast.receive called on the source rank")
        if self.manifest is None:
            box: list[Any] = [None]
            dist.broadcast_object_list(box, src=self.src, group=self.group)
            self._set_manifest([tuple(m) for m in box[0]])
        header = torch.empty(2, dtype=torch.int64, device=self.device)
        dist.broadcast(header, src=self.src, group=self.group)
        step, nbytes = (int(x) for x in header.tolist())
        if step < 0:
            return None
        if nbytes != self.total_bytes:
            raise RuntimeError(f"WeightBroadcast: sender announces {nbytes} bytes, manifest says {self.total_bytes}")
        bucket = self._buckets[self._turn]
        self._turn ^= 1
        dist.broadcast(bucket, src=self.src, group=self.group)
        return step, self._views(bucket)
