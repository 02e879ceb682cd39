"""Training step of the LGD path: two optimizers, distill-flag schedule, student-backbone freeze,
RCCL data parallelism  [ref: train.py:148-234 do_train, 279-281 DDP wrap; utils/build.py:492-553].

MI355X-first differences (behaviour-preserving):
  * no per-iteration host sync: the reference reduces and `.item()`s every loss each step
    (train.py:196); here losses stay on the device and are fetched once per log period, already
    averaged over ranks by ONE small all-reduce;
  * the backbone freeze is applied as `requires_grad=False` for the phase instead of computing,
    all-reducing and then discarding the gradients (train.py:205-207) -- SGD skips the same
    parameters (grad None), the backward pass and the gradient buckets shrink;
  * parameters that can never receive a gradient (global_ctx_proj_1D without a context box) are
    frozen statically, so DDP runs with find_unused_parameters=False (no per-iteration graph walk);
    `static_graph` stays off: the distill-flag switch changes which edges reach the backbone;
  * works at world size 1 (the reference dereferences `model.module` unconditionally).
"""
import bisect
import math
import os

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


def warmup_multistep_factor(it, steps, gamma, warmup_factor, warmup_iters, warmup_method="linear"):
    """detectron2 WarmupMultiStepLR multiplier at iteration `it` ([d2-memory])."""
    w = 1.0
    if it < warmup_iters:
        if warmup_method == "constant":
            w = warmup_factor
        elif warmup_method == "linear":
            a = it / warmup_iters
            w = warmup_factor * (1 - a) + a
        else:
            raise ValueError("Unknown warmup method: {}".format(warmup_method))
    return w * gamma ** bisect.bisect_right(list(steps), it)


def _warmup_factor(it, warmup_factor, warmup_iters, warmup_method):
    if it >= warmup_iters:
        return 1.0
    if warmup_method == "constant":
        return warmup_factor
    if warmup_method == "linear":
        a = it / warmup_iters
        return warmup_factor * (1 - a) + a
    raise ValueError("Unknown warmup method: {}".format(warmup_method))


def warmup_cosine_factor(it, max_iters, warmup_factor, warmup_iters, warmup_method="linear"):
    """detectron2 WarmupCosineLR multiplier ([d2-memory]): warmup * 0.5 * (1 + cos(pi * it / max_iters))."""
    return _warmup_factor(it, warmup_factor, warmup_iters, warmup_method) * 0.5 * (1.0 + math.cos(math.pi * it / max_iters))


def build_distillator_lr_scheduler(solver, optimizer, max_iter=None):
    """[ref: utils/build.py:531-553]; `max_iter` backs WarmupCosineLR when the sub-solver has no MAX_ITER of its own
    (the reference reads solver.MAX_ITER, which its default sub-config does not define)."""
    name = solver.LR_SCHEDULER_NAME
    if name == "WarmupMultiStepLR":
        f = lambda it: warmup_multistep_factor(it, solver.STEPS, solver.GAMMA, solver.WARMUP_FACTOR,  # noqa: E731
                                               solver.WARMUP_ITERS, solver.WARMUP_METHOD)
        return torch.optim.lr_scheduler.LambdaLR(optimizer, f)
    if name == "WarmupCosineLR":
        total = getattr(solver, "MAX_ITER", None) or max_iter
        if not total:
            raise ValueError("WarmupCosineLR needs MAX_ITER")
        f = lambda it: warmup_cosine_factor(it, total, solver.WARMUP_FACTOR, solver.WARMUP_ITERS, solver.WARMUP_METHOD)  # noqa: E731
        return torch.optim.lr_scheduler.LambdaLR(optimizer, f)
    raise ValueError("Unknown LR sheduler: {}".format(name))


def load_scheduler_state(scheduler, state):
    """restore a LambdaLR of build_distillator_lr_scheduler from its own state_dict OR from the state of the reference's
    detectron2 WarmupMultiStepLR / WarmupCosineLR (`milestones, gamma, warmup_*, base_lrs, last_epoch, _step_count, _last_lr`:
    no `lr_lambdas`, which LambdaLR.load_state_dict pops unconditionally; and `_last_lr` / `base_lrs` hold one entry per
    parameter, utils/build.py:494-512).  The schedule itself is a pure function of the iteration, so only the position is
    taken and the learning rates are recomputed for this optimizer's groups."""
    last = int(state["last_epoch"])
    scheduler.last_epoch = last
    scheduler._step_count = int(state.get("_step_count", last + 1))
    lrs = [base * lmbda(last) for base, lmbda in zip(scheduler.base_lrs, scheduler.lr_lambdas)]
    for g, lr in zip(scheduler.optimizer.param_groups, lrs):
        g["lr"] = lr
    scheduler._last_lr = lrs


def _unwrap(model):
    return model.module if isinstance(model, DistributedDataParallel) else model


def _params(modules):
    seen, out = set(), []
    for m in modules:
        for p in m.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def build_distillator_optimizer(cfg, network):
    """student+adapter share one optimizer, the teacher has its own; weight decay on every
    parameter, SGD momentum  [ref: utils/build.py:492-529].  Multi-tensor (`foreach`) updates."""
    net = _unwrap(network)
    d = cfg.MODEL.DISTILLATOR

    def make(solver, params):
        if solver.OPTIMIZER == "SGD":
            return torch.optim.SGD(params, solver.BASE_LR, momentum=solver.MOMENTUM, weight_decay=solver.WEIGHT_DECAY,
                                   foreach=True)
        if solver.OPTIMIZER == "ADAMW":
            return torch.optim.AdamW(params, solver.BASE_LR, betas=(0.9, 0.999), weight_decay=solver.WEIGHT_DECAY,
                                     foreach=True)
        raise NotImplementedError("no optimizer type %s" % solver.OPTIMIZER)
    # the reference registers every parameter that requires grad at build time -- including the ones that never receive a
    # gradient (global_ctx_proj_1D without a context box; SGD skips grad-less params) and the phase-frozen backbone -- in
    # named_parameters order, one param group each (utils/build.py:494-512).  Same parameters, same order here, but ONE group
    # per optimizer so that the multi-tensor update is a handful of launches instead of ~10 per parameter;
    # Trainer.state_dict()/load_state_dict() convert to / from the reference's one-group-per-parameter layout.
    return make(d.STUDENT.SOLVER, _all_params([net.student, net.adapter])), make(d.TEACHER.SOLVER, _all_params([net.teacher]))


def _all_params(modules):
    seen, out = set(), []
    for m in modules:
        for p in m.parameters():
            if id(p) not in seen and (p.requires_grad or getattr(p, "_lgd_phase_frozen", False)
                                      or getattr(p, "_lgd_never_trained", False)):
                seen.add(id(p))
                out.append(p)
    return out


def optimizer_state_to_reference(sd):
    """one param group per parameter, as the reference's optimizers store it (utils/build.py:494-512)."""
    out = {"state": sd["state"], "param_groups": []}
    for g in sd["param_groups"]:
        for i in g["params"]:
            out["param_groups"].append({**{k: v for k, v in g.items() if k != "params"}, "params": [i]})
    return out


def optimizer_state_from_reference(sd, optimizer):
    """merge a one-group-per-parameter state_dict into this optimizer's group layout (hyper-parameters of the first group:
    the reference gives every parameter the same lr / weight decay)."""
    mine = optimizer.state_dict()["param_groups"]
    if len(sd["param_groups"]) == len(mine):
        return sd
    flat = [i for g in sd["param_groups"] for i in g["params"]]
    if len(mine) != 1 or len(flat) != len(mine[0]["params"]):
        raise ValueError("optimizer state has %d parameters in %d groups, expected %d parameters"
                         % (len(flat), len(sd["param_groups"]), sum(len(g["params"]) for g in mine)))
    g0 = sd["param_groups"][0]
    merged = {**mine[0], **{k: v for k, v in g0.items() if k != "params"}, "params": flat}
    return {"state": sd["state"], "param_groups": [merged]}


def freeze_static_unused(model):
    """Parameters that never get a gradient in this configuration (reference: DDP find_unused_parameters)."""
    net = _unwrap(model)
    t = net.teacher
    unused = []
    if not t.add_context_box:
        unused += list(t.global_ctx_proj_1D.parameters())  # dynamic_teacher.py:153-190 never calls it
    if t.interact_pattern in ("student_fill", "teacher_fill"):
        unused += list(t.multi_head_attn.parameters())
    for p in unused:
        p.requires_grad = False
        p._lgd_never_trained = True
    return unused


class Trainer:
    """Owns model (+DDP), optimizers, schedulers and the per-iteration phase logic of do_train."""

    def __init__(self, cfg, model, distributed=None, device=None, fused_sgd=True):
        self.cfg = cfg
        self.d = cfg.MODEL.DISTILLATOR
        self.max_iter = cfg.SOLVER.MAX_ITER
        self.raw_model = model
        self.device = device or next(model.parameters()).device
        self.distributed = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) \
            if distributed is None else distributed
        freeze_static_unused(model)
        self._backbone_frozen = None
        self.model = model
        self._set_backbone_frozen(0 < self.d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)
        self.stu_optimizer, self.tea_optimizer = build_distillator_optimizer(cfg, model)
        self.stu_scheduler = build_distillator_lr_scheduler(self.d.STUDENT.SOLVER, self.stu_optimizer, self.max_iter)
        self.tea_scheduler = build_distillator_lr_scheduler(self.d.TEACHER.SOLVER, self.tea_optimizer, self.max_iter)
        self.clip = cfg.SOLVER.CLIP_GRADIENTS
        self.iteration = 0
        self._log_acc, self._log_n = None, 0
        self._finite = None  # device-side AND of isfinite(total loss) over the steps since the last check
        self._fused_sgd = None
        # one-launch filter folds of the student's trainable 1x1 convolutions (student/resnet.py::StepFolds); LGD_STEP_FOLDS=0: per-op folds
        self._step_folds = None
        if os.environ.get("LGD_STEP_FOLDS", "1") != "0" and next(self.raw_model.parameters()).is_cuda:
            from .student.resnet import StepFolds
            self._step_folds = StepFolds(self.raw_model)
        self.fused_sgd_enabled = fused_sgd   # False: torch's multi-tensor clip + SGD path (what the fused launch is tested against)
        if self.device.type == "cuda":
            from . import ops, optim
            self.tuned_gemms = ops.enable_tuned_gemms()  # opt-in here (not an import side effect): lookup-only solution table
            # clip + both SGD updates as one HIP launch (csrc/optim.hip); other optimizers / clip types: torch's multi-tensor path
            if self.fused_sgd_enabled and optim.supported([self.stu_optimizer, self.tea_optimizer], self.clip):
                self._fused_sgd = optim.FusedClipSGD([self.stu_optimizer, self.tea_optimizer],
                                                     self.clip.CLIP_VALUE if self.clip.ENABLED else None)
        else:
            self.tuned_gemms = False

    # ---- phases ----------------------------------------------------------------------------
    def _set_backbone_frozen(self, frozen):
        """[ref: train.py:205-207] as a requires_grad phase; the DDP reducer is rebuilt at the (rare) phase changes."""
        if frozen == self._backbone_frozen:
            return
        net = self.raw_model
        for p in net.student.raw_backbone.parameters():
            if getattr(p, "_lgd_trainable", None) is None:
                p._lgd_trainable = p.requires_grad  # FREEZE_AT already froze stem/res2 for good
            if p._lgd_trainable:
                p.requires_grad = not frozen
                p._lgd_phase_frozen = frozen
                if frozen:
                    p.grad = None
        self._backbone_frozen = frozen
        if self.distributed:
            dev_ids = [self.device.index] if self.device.type == "cuda" else None
            self.model = DistributedDataParallel(net, device_ids=dev_ids, broadcast_buffers=False,
                                                 find_unused_parameters=False, gradient_as_bucket_view=True,
                                                 bucket_cap_mb=int(os.environ.get("LGD_BUCKET_MB", "32")))
            # 32 MB: the bucket that becomes ready LAST does so 0.07 ms before backward ends on both BASELINE shapes (tools/
            # ddp_bucket_times.py, profiles/r03_ddp_world1_ab_and_bucket_times.txt), so its all-reduce is the part of the exchange
            # no compute can hide and its size is what an N-rank step pays; the earlier buckets have 5-15 ms of backward left.
            # (LGD_BUCKET_MB: deployment knob for other fabrics.)
        else:
            self.model = net

    def set_phase(self, iteration):
        """distill flag and backbone freeze for this iteration [ref: train.py:184-189,205-207]."""
        d = self.d
        if iteration < d.PRE_NONDISTILL_ITERS or iteration > self.max_iter - d.POST_NONDISTILL_ITERS:
            self.raw_model.distill_flag = d.DISTILL_OFF
        else:
            self.raw_model.distill_flag = d.DISTILL_ON
        self._set_backbone_frozen(iteration < d.PRE_FREEZE_STUDENT_BACKBONE_ITERS)

    # ---- one iteration ----------------------------------------------------------------------
    def step(self, data, iteration=None):
        it = self.iteration if iteration is None else iteration
        self.set_phase(it)
        # nn.Module.train() walks the whole module tree: once, not per step.  Both flags: an evaluation in the middle of training
        # (the reference runs do_test inside the loop) calls raw_model.eval(), which leaves a DDP wrapper's own flag at True
        if not (self.model.training and self.raw_model.training):
            self.model.train()
        if self._step_folds is not None:
            self._step_folds.prepare()   # w * scale of every trainable 1x1 ConvBN of the student: one launch per step
        loss_dict = self.model(data)
        losses = sum(loss_dict.values())
        ok = torch.isfinite(losses.detach())  # the reference asserts this every iteration (train.py:194); here: no host sync
        self._finite = ok if self._finite is None else self._finite & ok
        if self._fused_sgd is not None:
            self._fused_sgd.zero_grad()
            losses.backward()  # DDP: bucketed RCCL all-reduce over xGMI overlaps with this
            self._fused_sgd.step()   # clip + stu_optimizer.step() + tea_optimizer.step(): one launch
        else:
            self.stu_optimizer.zero_grad(set_to_none=True)
            self.tea_optimizer.zero_grad(set_to_none=True)
            losses.backward()
            if self.clip.ENABLED:
                self._clip()
            self.stu_optimizer.step()
            self.tea_optimizer.step()
        self.stu_scheduler.step()
        self.tea_scheduler.step()
        # device-side running sums; no host sync here
        vals = torch.stack([v.detach() for v in loss_dict.values()])
        self._log_keys = list(loss_dict.keys())
        self._log_acc = vals if self._log_acc is None else self._log_acc + vals
        self._log_n += 1
        self.iteration = it + 1
        return loss_dict

    def _clip(self):
        """detectron2 maybe_add_gradient_clipping: per PARAMETER, CLIP_TYPE 'value' (default, CLIP_VALUE 1.0) = element-wise
        clamp, 'norm' = clip_grad_norm_ of each parameter on its own [d2-memory]; multi-tensor forms, no host sync."""
        grads = [p.grad for g in self.stu_optimizer.param_groups + self.tea_optimizer.param_groups
                 for p in g["params"] if p.grad is not None]
        if not grads:
            return
        v = float(self.clip.CLIP_VALUE)
        if self.clip.CLIP_TYPE == "value":
            torch._foreach_clamp_min_(grads, -v)
            torch._foreach_clamp_max_(grads, v)
        else:
            norms = torch._foreach_norm(grads, float(self.clip.NORM_TYPE))
            coef = [torch.clamp(v / (n + 1e-6), max=1.0) for n in norms]
            torch._foreach_mul_(grads, coef)

    def check_finite(self):
        """raise if any total loss since the last check was non-finite (ONE host sync; call before saving a checkpoint)."""
        if self._finite is None:
            return
        if self.distributed:
            f = self._finite.to(torch.float32)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            ok = bool(f.item() > 0)
        else:
            ok = bool(self._finite.item())
        self._finite = None
        if not ok:
            raise FloatingPointError("non-finite total loss at or before iteration %d" % (self.iteration - 1))

    def fetch_metrics(self):
        """Mean of every loss since the last call, averaged over ranks (ONE all-reduce, ONE host copy),
        plus total_loss / stu_lr / tea_lr  [ref: train.py:196-199, 212-213 scalar names].  Raises if a loss went
        non-finite (the reference asserts every iteration, train.py:194)."""
        self.check_finite()
        if self._log_acc is None:
            return {}
        v = self._log_acc / self._log_n
        if self.distributed:
            dist.all_reduce(v)
            v = v / dist.get_world_size()
        vals = v.tolist()
        self._log_acc, self._log_n = None, 0
        out = dict(zip(self._log_keys, vals))
        out["total_loss"] = sum(vals)
        out["stu_lr"] = self.stu_optimizer.param_groups[0]["lr"]
        out["tea_lr"] = self.tea_optimizer.param_groups[0]["lr"]
        if not all(x == x and abs(x) != float("inf") for x in vals):
            raise FloatingPointError("non-finite loss: %s" % out)
        return out

    def state_dict(self):
        """checkpoint payload with the reference's keys and optimizer layout [ref: train.py:155-157, utils/build.py:494-512]."""
        return {"model": self.raw_model.state_dict(),
                "stu_optimizer": optimizer_state_to_reference(self.stu_optimizer.state_dict()),
                "tea_optimizer": optimizer_state_to_reference(self.tea_optimizer.state_dict()),
                "stu_scheduler": self.stu_scheduler.state_dict(),
                "tea_scheduler": self.tea_scheduler.state_dict(), "iteration": self.iteration - 1}

    def load_state_dict(self, sd):
        self.raw_model.load_state_dict(sd["model"])
        self.stu_optimizer.load_state_dict(optimizer_state_from_reference(sd["stu_optimizer"], self.stu_optimizer))
        self.tea_optimizer.load_state_dict(optimizer_state_from_reference(sd["tea_optimizer"], self.tea_optimizer))
        load_scheduler_state(self.stu_scheduler, sd["stu_scheduler"])
        load_scheduler_state(self.tea_scheduler, sd["tea_scheduler"])
        self.iteration = sd.get("iteration", -1) + 1
