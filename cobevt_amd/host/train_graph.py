"""One optimisation step of the reference's training loop (opv2v/opencood/tools/train_camera.py:143-179: zero_grad, forward, criterion,
backward, optimizer.step) captured into a HIP graph and replayed - for the host-bound regime (few agents: the eager step is ~3000
launches).  The training forward / backward was built for this: no host -> device copies and no host syncs (pose algebra from fills + the
device inverse, class weights uploaded once, the label-range check deferred), a device word in the attention-dropout seed
(autograd.dropout_step, bumped inside the graph so that every replay draws new masks), detached VanillaSegLoss.loss_dict entries (a kept
loss tensor pins the step's autograd graph, and on ROCm 7.2 kills hipGraphInstantiate when the pinned gradient accumulators live on the
default stream).  tools/train_graph_probe.py times it against the eager step."""
import warnings

import torch

from .. import autograd as ag
from ..lib import CobevtHipError
from .runtime import GraphOwner


def _state_tensors(optimizer):
    for st in optimizer.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                yield v


class CapturedTrainStep(GraphOwner):
    """step(batch) == one eager training step on `batch`, returning the (static) loss tensor.

    model      a HipModule in train() mode on a ROCm device (CorpBEVT / FaxFusedTransformer)
    criterion  callable (output_dict, batch) -> scalar loss (e.g. VanillaSegLoss via a lambda picking the ground-truth keys)
    optimizer  torch.optim.SGD, or Adam / AdamW built with capturable=True
    example_batch  dict of tensors with the shapes every later batch will have
    reducer    optional cobevt_amd.dist.GradAllReducer over the same parameters (world size > 1)
    warmup     eager steps run before the capture (they build the optimizer state and the autograd / allocator caches); the
               model, its buffers and the optimizer state are restored afterwards, so the capture starts from the given state
    autocast_dtype  torch.bfloat16: forward + criterion inside torch.autocast (train_camera.py --half with bf16; no GradScaler is
               needed for bf16); None: fp32

    What to expect from a replay: the same kernels on the same data as the eager step, so the same numbers up to the order of the
    fp32 atomics (weight gradients, bias sums) - and, as between any two runs of a deep ReLU network, the occasional pre-activation
    that lands on the other side of zero (one flipped ReLU moves one row of one weight gradient by up to ~1 % of that
    tensor's scale; tests/test_training_gpu.py::test_captured_train_step_follows_eager compares K replayed steps with K eager ones).
    When it pays: the eager step issues ~3000 launches; with 2 agents that is more host time than GPU time.
    """

    def __init__(self, model, criterion, optimizer, example_batch, reducer=None, warmup=2, autocast_dtype=None):
        if not model.training:
            raise CobevtHipError("CapturedTrainStep captures a training step: call model.train() first")
        p0 = next(model.parameters())
        if not p0.is_cuda:
            raise CobevtHipError("CapturedTrainStep needs the model on a ROCm device")
        if isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)) and not optimizer.defaults.get("capturable", False):
            raise CobevtHipError("build Adam / AdamW with capturable=True: their step counters must live on the device to be replayed")
        self.model, self.criterion, self.optimizer, self.reducer = model, criterion, optimizer, reducer
        self.autocast_dtype = autocast_dtype
        self.device = p0.device
        self.static_batch = {k: (v.to(self.device).clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        if "record_len" in example_batch:          # the agent counts shape the graph: read them on the host once, before the capture
            self.static_batch["record_len_host"] = [int(v) for v in example_batch["record_len"]]
        self.loss = None
        self.graphs = None
        self._warm_up(warmup)
        self._capture()

    # ------------------------------------------------------------------------------------------------------------------
    def _forward_backward(self):
        self.optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=self.autocast_dtype or torch.bfloat16, enabled=self.autocast_dtype is not None):
            loss = self.criterion(self.model(dict(self.static_batch)), self.static_batch)
        loss.backward()
        return loss.detach()

    def _eager(self):
        if self.reducer is not None:
            self.reducer.enabled = False
        loss = self._forward_backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()
        return loss

    def _warm_up(self, n):
        saved = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        # optimizer state that already exists (resumed momentum / Adam moments and step, eager steps taken before the capture)
        # is put back afterwards; state the warm-up creates is zeroed = the freshly built state
        saved_opt = {id(t): t.detach().clone() for t in _state_tensors(self.optimizer)}
        # warm-up and capture share ONE side stream: a parameter's gradient accumulator is bound to the stream it was created on,
        # and a capture that has to reach an accumulator living on the default stream dies inside hipGraphInstantiate.  That happens
        # when the autograd graph of an EARLIER eager step is still referenced (a kept loss tensor pins the accumulators it was
        # built with); torch warns about exactly this mismatch during the warm-up backward, which is turned into an error here.
        self._stream = side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.cuda.stream(side), torch.enable_grad():
                for _ in range(max(1, n)):
                    self._eager()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        with torch.no_grad():
            cur = self.model.state_dict()
            for k, v in saved.items():
                cur[k].copy_(v)
            for t in _state_tensors(self.optimizer):
                old = saved_opt.get(id(t))
                if old is not None:
                    t.copy_(old)
                else:                                      # zero = the freshly built state of SGD(momentum) / Adam / AdamW
                    t.zero_()
        self.optimizer.zero_grad(set_to_none=True)
        for w in caught:
            if "AccumulateGrad node's stream does not match" in str(w.message):
                raise CobevtHipError("CapturedTrainStep: a parameter's gradient accumulator is pinned to another stream - an output / "
                                     "loss tensor of an earlier eager step is still referenced (e.g. kept in a list or a logger); drop "
                                     "those references (or detach what you keep) before capturing the step")

    def _capture(self):
        dev = self.device
        step_word = ag.dropout_step(dev)
        ag.begin_capture_zero_pool()          # zero-initialised gradient buffers: slices of chunks whose fills are nodes of these graphs
        try:
            self._capture_graphs(step_word)
        finally:
            ag.end_capture_zero_pool()

    def _capture_graphs(self, step_word):
        if self.reducer is None:
            g = torch.cuda.CUDAGraph()
            with torch.enable_grad(), torch.cuda.graph(g, stream=self._stream):
                self.loss = self._forward_backward()
                self.optimizer.step()
                step_word.add_(1)
            self.graphs = (g,)
        else:
            self.reducer.enabled = False               # the all-reduce runs between the graphs (finish() issues every bucket)
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.enable_grad(), torch.cuda.graph(g1, stream=self._stream):
                self.loss = self._forward_backward()
            with torch.cuda.graph(g2, pool=g1.pool(), stream=self._stream):
                self.optimizer.step()
                step_word.add_(1)
            self.graphs = (g1, g2)
        # the capture itself did not run anything: parameters, optimizer state and BatchNorm statistics are still the restored ones

    # ------------------------------------------------------------------------------------------------------------------
    def load(self, batch):
        """copy a new batch into the static buffers (shapes fixed at capture)"""
        for k, v in batch.items():
            dst = self.static_batch.get(k)
            if k == "record_len" and torch.is_tensor(v) and not v.is_cuda \
                    and [int(x) for x in v] != self.static_batch["record_len_host"]:
                raise CobevtHipError("CapturedTrainStep captured record_len %s, got %s (capture a new step per agent-count pattern)"
                                     % (self.static_batch["record_len_host"], [int(x) for x in v]))
            if not torch.is_tensor(dst) or not torch.is_tensor(v):
                continue
            if tuple(v.shape) != tuple(dst.shape):
                raise CobevtHipError("CapturedTrainStep captured %s of shape %s, got %s (capture a new step for a new shape)"
                                     % (k, tuple(dst.shape), tuple(v.shape)))
            dst.copy_(v, non_blocking=True)

    def step(self, batch=None):
        if batch is not None:
            self.load(batch)
        self.graphs[0].replay()
        if self.reducer is not None:
            self.reducer.finish()
            self.graphs[1].replay()
        return self.loss
