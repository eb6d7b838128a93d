"""Training of SuperResolutionAE on libsrcfd (sr-ae-conv.ipynb:c289-320, c546-560).

`Trainer.step(x_lr, x_hr)` = one `train_step`: loss = mean over all elements of (x_hr - pred)^2,
gradients for every trainable weight, Keras-default Adam.  Data parallel over the GPUs of one
node: one process per GPU, identical replicas, every rank runs forward/backward on its own
micro-batch with the loss normalised by the GLOBAL batch, then ONE all-reduce(sum) over the flat
2 709 491-float gradient buffer (RCCL through `torch.distributed`, backend "nccl"), then the same
Adam update everywhere.  torch is used for device buffers and the collective only.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, List, Optional

import numpy as np

from . import _lib as L
from .engine import SRModel


def allreduce_sum_(flat_grads, async_op: bool = False):
    """The step's only collective: in-place SUM over ranks of the flat gradient tensor.
    async_op=True returns the collective's work handle (None with one rank): with the RCCL backend the reduction runs on
    the backend's own stream behind an event of the caller's stream, and `handle.wait()` makes the caller's STREAM wait
    for it, not the host -- whatever the host enqueues in between (the next batch's staging) is not held up."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, async_op=async_op) if async_op else dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return None


def world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def epoch_batches(n_samples: int, batch_size: int, epoch: int, seed: int, rank_: int = 0, world: int = 1) -> Iterator[np.ndarray]:
    """`Dataset.shuffle(len).batch(batch_size)` reshuffled every epoch (sr-ae-conv.ipynb:c558): a
    seeded permutation per epoch, cut into global batches of batch_size*world; each rank takes its
    contiguous slice of every global batch (the last one may be ragged, like Keras')."""
    for idx, _ in epoch_batches_global(n_samples, batch_size, epoch, seed, rank_, world):
        yield idx


def epoch_batches_global(n_samples: int, batch_size: int, epoch: int, seed: int, rank_: int = 0, world: int = 1):
    """As `epoch_batches`, yielding (this rank's indices, size of the global batch): every rank can
    compute the loss normalisation without a collective."""
    perm = np.random.default_rng([seed, epoch]).permutation(n_samples)
    gb = batch_size * world
    for lo in range(0, n_samples, gb):
        idx = perm[lo:lo + gb]
        per = -(-len(idx) // world)
        yield idx[rank_ * per:(rank_ + 1) * per], len(idx)


class Trainer:
    def __init__(self, model: SRModel, max_batch: int = 8, lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999,
                 eps: float = 1e-7):
        import torch
        if model.device < 0:
            raise L.NoDeviceError("training needs a device handle")
        self.model = model
        self.device = torch.device("cuda", model.device)
        self._h = C.c_void_p()
        L.check(L.lib.srcfd_trainer_create(model._h, int(max_batch), C.byref(self._h)))
        self.n_params = int(L.lib.srcfd_trainer_num_params(self._h))
        host = np.empty(self.n_params, np.float32)
        L.check(L.lib.srcfd_trainer_get_params(self._h, host.ctypes.data_as(C.c_void_p)))
        self.params = torch.from_numpy(host).to(self.device)
        self.grads = torch.zeros_like(self.params)
        self.m = torch.zeros_like(self.params)
        self.v = torch.zeros_like(self.params)
        self.sse = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.max_batch = max_batch
        oh, ow, oc = model.output_shape
        self.out_elems = oh * ow * oc
        self._layout = [(d["name"], d["kernel"].shape, d["bias"].shape) for d in model.layers() if "kernel" in d]

    def close(self):
        if getattr(self, "_h", None):
            L.lib.srcfd_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward_backward(self, x, y, global_batch: Optional[int] = None, overwrite: bool = False, same_params: bool = False) -> None:
        """Accumulates this rank's gradient of the GLOBAL mean-squared error into self.grads (and the squared-error sum into
        self.sse); overwrite=True stores them instead (SRCFD_TRAIN_OVERWRITE: no zero-fill launches in front of the step).
        same_params=True: self.params has not changed since this trainer's previous call (later micro-batches of one
        optimiser step) -- the re-packing of the parameters is skipped (SRCFD_TRAIN_SAME_PARAMS)."""
        import torch
        n = int(x.shape[0])
        gb = global_batch if global_batch is not None else n
        st = torch.cuda.current_stream(self.device)
        L.check(L.lib.srcfd_trainer_forward_backward_ex(
            self._h, C.c_void_p(self.params.data_ptr()), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), n,
            C.c_float(1.0 / (gb * self.out_elems)), C.c_void_p(self.grads.data_ptr()), C.c_void_p(self.sse.data_ptr()),
            (L.TRAIN_OVERWRITE if overwrite else 0) | (L.TRAIN_SAME_PARAMS if same_params else 0), C.c_void_p(st.cuda_stream)))

    def apply_adam(self) -> None:
        import torch
        self.t += 1
        st = torch.cuda.current_stream(self.device)
        L.check(L.lib.srcfd_adam_step(C.c_void_p(self.params.data_ptr()), C.c_void_p(self.grads.data_ptr()), C.c_void_p(self.m.data_ptr()),
                                      C.c_void_p(self.v.data_ptr()), self.n_params, self.t, C.c_float(self.lr), C.c_float(self.beta1),
                                      C.c_float(self.beta2), C.c_float(self.eps), C.c_void_p(st.cuda_stream)))

    def step(self, x, y, global_batch: Optional[int] = None, return_loss: bool = True):
        """One optimisation step on this rank's micro-batch (contiguous float32 CUDA tensors).  Returns
        the global mean-squared error of the batch (like Keras' `recon_loss`); with return_loss=False
        nothing is read back (no host synchronisation) and the squared-error sum stays in `self.sse`.
        global_batch: samples of the step over ALL ranks (the loss normalisation).  None = this rank's count x the world
        size, i.e. equal micro-batches; a ragged global batch (Keras' last batch of an epoch) must be passed explicitly --
        every rank knows it from `epoch_batches_global` without a collective (no per-step all-reduce + host read-back)."""
        import torch.distributed as dist
        n = int(x.shape[0])
        w = world_size()
        if global_batch is None:
            global_batch = n * w
        if n:
            self.forward_backward(x, y, global_batch, overwrite=True)   # stores grads and sse: nothing to zero
        else:                                                           # a rank without samples in a ragged last batch
            self.grads.zero_()
            self.sse.zero_()
        work = allreduce_sum_(self.grads, async_op=True)
        if work is not None:
            work.wait()          # stream-side wait (RCCL): Adam follows the reduction, the host runs on
        self.apply_adam()
        if not return_loss:
            return None
        if w > 1:
            dist.all_reduce(self.sse)
        return float(self.sse.item()) / (global_batch * self.out_elems)

    # -- weights in / out -------------------------------------------------------
    def weights(self) -> Dict[str, np.ndarray]:
        flat = self.params.detach().cpu().numpy()
        out, off = {}, 0
        for name, ks, bs in self._layout:
            kn, bn = int(np.prod(ks)), int(np.prod(bs))
            out[f"{name}/kernel"] = flat[off:off + kn].reshape(ks).copy(); off += kn
            out[f"{name}/bias"] = flat[off:off + bn].reshape(bs).copy(); off += bn
        assert off == self.n_params
        return out

    def export_model(self, device: Optional[int] = None) -> SRModel:
        """A fresh inference handle with the trained weights (same layer graph)."""
        w = self.weights()
        specs = []
        kinds = {L.LAYER_CONV2D: "conv2d", L.LAYER_CONV2D_TRANSPOSE: "conv2d_transpose", L.LAYER_DENSE: "dense",
                 L.LAYER_FLATTEN: "flatten", L.LAYER_RESHAPE: "reshape"}
        acts = {L.ACT_LINEAR: "linear", L.ACT_SWISH: "swish", L.ACT_RELU: "relu", L.ACT_SIGMOID: "sigmoid", L.ACT_TANH: "tanh"}
        for d in self.model.layers():
            s = dict(kind=kinds[d["kind"]], name=d["name"], k=d["kh"], stride=d["stride"], same=d["same"], act=acts[d["activation"]])
            if d["kind"] == L.LAYER_RESHAPE:
                s["shape"] = d["reshape"]
            if "kernel" in d:
                s["w"], s["b"] = w[f"{d['name']}/kernel"], w[f"{d['name']}/bias"]
            specs.append(s)
        return SRModel.from_layers(specs, self.model.input_shape, self.model.device if device is None else device)


def fit(trainer: Trainer, x_lr: np.ndarray, x_hr: np.ndarray, epochs: int, batch_size: int = 8, seed: int = 0,
        log_every: int = 0) -> List[float]:
    """`model.fit(ds, epochs=...)` (sr-ae-conv.ipynb:c558-560) for host arrays; returns per-epoch mean loss."""
    import torch
    r, w = rank(), world_size()
    xs = torch.from_numpy(np.ascontiguousarray(x_lr, np.float32)).to(trainer.device)
    ys = torch.from_numpy(np.ascontiguousarray(x_hr, np.float32)).to(trainer.device)
    history = []
    epoch_loss = torch.zeros(1, dtype=torch.float64, device=trainer.device)
    # Input pipeline (`shuffle(len).batch(8)`, c558): the gather of batch k + 1 runs on a side stream while step k (kernels,
    # gradient all-reduce, Adam) runs on the main one; the main stream waits for a batch's event only, never the host.
    main = torch.cuda.current_stream(trainer.device)
    side = torch.cuda.Stream(device=trainer.device)
    side.wait_stream(main)       # xs / ys were uploaded on the main stream

    def gather(idx):
        with torch.cuda.stream(side):
            sel = torch.from_numpy(np.ascontiguousarray(idx)).to(trainer.device, non_blocking=True)
            xb, yb = xs.index_select(0, sel), ys.index_select(0, sel)
            ev = torch.cuda.Event()
            ev.record(side)
        return xb, yb, ev

    for ep in range(epochs):
        epoch_loss.zero_()
        steps = 0
        batches = list(epoch_batches_global(len(xs), batch_size, ep, seed, r, w))
        nxt = gather(batches[0][0]) if batches else None
        for i, (idx, gb) in enumerate(batches):
            xb, yb, ev = nxt
            if i + 1 < len(batches):
                nxt = gather(batches[i + 1][0])
            main.wait_event(ev)
            xb.record_stream(main)
            yb.record_stream(main)
            trainer.step(xb, yb, global_batch=gb, return_loss=False)
            epoch_loss += trainer.sse / (gb * trainer.out_elems)   # this rank's share of the batch loss, on the device
            steps += 1
        if w > 1:
            import torch.distributed as dist
            dist.all_reduce(epoch_loss)
        history.append(float(epoch_loss.item()) / max(steps, 1))   # one read-back per epoch: mean of the batch losses
        if log_every and r == 0 and (ep + 1) % log_every == 0:
            print(f"epoch {ep + 1}: recon_loss {history[-1]:.6f}")
    return history
