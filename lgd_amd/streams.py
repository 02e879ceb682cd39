"""Second HIP streams for chains of the step that do not depend on each other (the teacher's label encoder beside the backbone, the box tower of
the head beside the class tower): one process per GPU still, the streams overlap tails and small launches of one chain with the other's kernels.
Autograd runs every backward node on the stream of its forward, so the overlap carries over to the backward pass.
[ref: the reference issues everything on one stream -- train.py:182-215; the chains themselves: thirdparty_heads/fcos.py:520-546,
 detectron2 RetinaNetHead.forward]"""
import os
import weakref

import torch

_SIDE = {}   # (device index, name) -> torch.cuda.Stream
_MAIN = {}   # device index -> the stream the step runs on (the one a fork was last taken from)


_SIDE_RAW = {}  # device index -> {raw handle of a side stream: the stream object}
_LIB_LAST = {}  # device index -> {raw handle of a side stream: event behind its last library call}
_LIB_SEEN = {}  # (device index, raw handle of a waiting stream, raw handle of a side stream) -> the event that stream has already waited for


# All forks of the step share ONE physical side stream per device (LGD_ONE_SIDE_STREAM=0: one per name, the round-5 form).  The forks are active in
# different phases of the step (label encoder under the backbone, FPN output convolution inside the FPN, adapter beside the teacher, class tower
# beside the box tower; the backward visits them in reverse), so one stream loses no overlap -- and the process then holds TWO streams, which HIP
# maps onto two of its 4 hardware queues whatever else the process created: with a stream per fork the fourth one shared the main stream's queue
# (its fork serialised: whichever fork came last gained nothing) and more than 4 queues stalled (profiles/r06_hw_queues_and_forks.txt).
_ONE_SIDE = os.environ.get("LGD_ONE_SIDE_STREAM", "1") != "0"


def _key(idx, name):
    return (idx, "side" if _ONE_SIDE else name)


def side(device, name):
    """the side stream a fork called `name` runs on (created on first use)"""
    key = _key(device.index if device.index is not None else torch.cuda.current_device(), name)
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = torch.cuda.Stream(device)
        _SIDE_RAW.setdefault(key[0], {})[s.cuda_stream] = s
    return s


_ORDER_LIBRARY = os.environ.get("LGD_LIBRARY_ORDER", "1") != "0"   # 0: the round-5 behaviour (tools/stall_repro.sh reproduces the stall with it)


class library_call:
    """`with streams.library_call(device):` around EVERY rocBLAS / MIOpen call this package issues (torch.bmm, baddbmm, F.conv2d and the
    convolution backward): calls on different streams are ORDERED, never concurrent.

    Root cause of round 5's two-stream stall (tools/stall_repro.sh, profiles/r06_stall_root_cause.txt): torch keeps ONE rocBLAS handle per host
    thread and points it at whatever stream is current (at::cuda::getCurrentCUDABlasHandle -> rocblas_set_stream); the handle owns ONE device
    workspace, which the library's split-K kernels (the long-K weight-gradient products dU = dM V^T of the F(4x4) variant, 36 batches) use for
    partial tiles and for the flags their workgroups spin on.  Two such GEMMs issued from two streams overlap on the device, one resets the other's
    flags, and its workgroups spin for ever: GPU 100 % busy, memory idle, every later launch of that stream queued behind it -- reproduced in
    isolation on the second step of the F(4x4) variant with only the head fork on, with and without the tuning table.  (CUDA builds of torch give
    every (handle, stream) pair its own workspace -- cublasSetWorkspace; the ROCm build does not.)  The shipped F(6x6) path keeps its large
    products on this library's own kernels (no workspace, no inter-workgroup waits), but bbox_pred's C' = 36 products, everything under the size
    gates at 2 images per GPU and the small FPN levels ARE library calls on side streams: the same hazard with better odds.
    Ordering costs nothing where it matters: the forks exist to overlap this library's launches, tails and small kernels, not two GEMMs of the
    vendor library (measured slower in round 3).  Calls on the step's main stream cost one dictionary probe; a call on a side stream waits for
    the main stream's work so far and for the other side streams' last library call, and leaves an event the others wait for."""
    __slots__ = ("idx", "cur", "raw")

    def __init__(self, device):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.cur = None

    def __enter__(self):
        sides = _SIDE_RAW.get(self.idx)
        if not sides or not _ORDER_LIBRARY:   # no fork has ever been taken on this device: one stream, stream order
            return self
        raw = self.raw = torch._C._cuda_getCurrentRawStream(self.idx)
        last = _LIB_LAST.get(self.idx)
        cur = sides.get(raw)
        if cur is not None:    # on a side stream
            self.cur = cur
            m = _MAIN.get(self.idx)
            if m is not None:
                cur.wait_stream(m)
            if last:
                for r2, ev in last.items():
                    if r2 != raw:
                        cur.wait_event(ev)
        elif last:             # on the main stream (or any other): behind the side streams' last library calls, each waited for once
            st = None
            for r2, ev in last.items():
                if _LIB_SEEN.get((self.idx, raw, r2)) is not ev:
                    if st is None:
                        st = torch.cuda.current_stream(self.idx)
                    st.wait_event(ev)
                    _LIB_SEEN[(self.idx, raw, r2)] = ev
        return self

    def __exit__(self, *exc):
        if self.cur is not None:
            ev = torch.cuda.Event()
            ev.record(self.cur)
            _LIB_LAST.setdefault(self.idx, {})[self.raw] = ev
        return False


def fork(device, name, inputs=()):
    """-> (main, side): the side stream waits for everything issued on the current stream so far; `inputs` (tensors made on the current stream
    that the side stream is about to read) are recorded on it, so that the allocator does not hand their memory out again under it"""
    main = torch.cuda.current_stream(device)
    _MAIN[main.device.index] = main
    s = side(device, name)
    s.wait_stream(main)
    for t in inputs:
        t.record_stream(s)
    return main, s


def record_all(obj, stream):
    """record_stream on every device tensor reachable from obj (tensors, nested lists / tuples / dicts) AND on the magnitude-tag word a tensor
    carries (ops._amax_tag: a slice of the zero-word pool of the stream that PRODUCED the tensor): all of them were allocated on one stream and are
    about to be read on `stream` -- the caching allocator must not hand their memory out again under it"""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
            tag = getattr(obj, "_lgd_amax", None)
            if tag is not None and isinstance(tag[0], torch.Tensor) and tag[0].is_cuda:
                tag[0].record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            record_all(o, stream)
    elif isinstance(obj, dict):
        for o in obj.values():
            record_all(o, stream)


def done(s):
    """an event behind what the side stream has been given so far: recorded where a fork's work ENDS, so that its join waits for that work only
    (the forks share one side stream: a later fork issued on it before this one is joined -- the adapter between the label encoder's fork and
    its join -- must not be waited for)"""
    ev = torch.cuda.Event()
    ev.record(s)
    return ev


def join(main, s, outputs=(), event=None):
    """the current stream waits for the side stream -- for `event` (done(s)) if given, else for everything issued on it so far; `outputs` (made on
    the side stream, read on the current one from here on: tensors or nested containers of them) are recorded on it together with their
    magnitude tags"""
    if event is not None:
        main.wait_event(event)
    else:
        main.wait_stream(s)
    record_all(outputs, main)


_HOOKED = {}   # id(parameter) -> (weak reference to it, names of the streams it is joined to): kept OUTSIDE the Parameter, whose __dict__ a
               # whole-model pickle / deepcopy carries along while the hooks themselves stay behind


def _multi_rank():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def join_on_grad(params, name):
    """Data-parallel runs: DistributedDataParallel starts a bucket's all-reduce from the gradient hook of the bucket's LAST parameter and orders it
    behind the stream THAT hook runs on.  Parameters whose gradients are written on a side stream get a post-accumulate hook that joins the two
    streams (each waits for the other's work so far), so whichever hook of a bucket comes last, its stream has seen every gradient of the bucket.
    Registered once per (parameter object, stream name), and only once the process group has more than one rank (called on every forward pass: a
    single-process run never pays for the hooks, a model that was copied or unpickled gets them on its first multi-rank forward)."""
    if not _multi_rank():
        return

    def hook(p):
        s, m = _SIDE.get(_key(p.device.index, name)), _MAIN.get(p.device.index)
        if s is not None and m is not None:
            s.wait_stream(m)
            m.wait_stream(s)
    for p in params:
        ent = _HOOKED.get(id(p))
        if ent is None or ent[0]() is not p:
            ent = _HOOKED[id(p)] = (weakref.ref(p, lambda _r, k=id(p): _HOOKED.pop(k, None)), set())
        if name not in ent[1]:
            p.register_post_accumulate_grad_hook(hook)
            ent[1].add(name)
