"""world-size-2 `gloo` tests of the N>1 path (CPU): DDP gradient averaging through lgd_amd.engine.Trainer,
phase switches that rebuild the reducer, rank-averaged metrics with one all-reduce, FCOS packed count all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Stub(nn.Module):
    """same attribute surface the engine touches: .student.raw_backbone, .adapter, .teacher, .distill_flag."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.student = nn.Module()
        self.student.raw_backbone = nn.Linear(4, 4)
        self.student.head = nn.Linear(4, 2)
        self.adapter = nn.ModuleDict({"distill": nn.Linear(4, 4)})
        self.teacher = nn.Module()
        self.teacher.add_context_box = False
        self.teacher.interact_pattern = "stuGuided"
        self.teacher.global_ctx_proj_1D = nn.Linear(4, 4)  # never used: must be frozen statically
        self.teacher.multi_head_attn = nn.Linear(4, 4)
        self.distill_flag = 0

    def forward(self, x):
        h = self.student.raw_backbone(x)
        stu = h if self.distill_flag else h.detach()
        return {"loss_cls": self.student.head(h).pow(2).mean(),
                "loss_distill": (self.adapter["distill"](stu) - self.teacher.multi_head_attn(x)).pow(2).mean()}


def _cfg():
    from lgd_amd import config
    return config.setup_cfg(None, ["MODEL.DEVICE", "cpu", "SOLVER.MAX_ITER", "100", "SOLVER.CLIP_GRADIENTS.ENABLED", "True",
                                   "MODEL.DISTILLATOR.PRE_NONDISTILL_ITERS", "4", "MODEL.DISTILLATOR.PRE_FREEZE_STUDENT_BACKBONE_ITERS", "2",
                                   "MODEL.DISTILLATOR.STUDENT.SOLVER.LR_SCHEDULER_NAME", "WarmupMultiStepLR", "MODEL.DISTILLATOR.STUDENT.SOLVER.STEPS", "(50,)",
                                   "MODEL.DISTILLATOR.STUDENT.SOLVER.GAMMA", "0.1", "MODEL.DISTILLATOR.STUDENT.SOLVER.WARMUP_FACTOR", "1.0",
                                   "MODEL.DISTILLATOR.STUDENT.SOLVER.WARMUP_ITERS", "0", "MODEL.DISTILLATOR.STUDENT.SOLVER.WARMUP_METHOD", "linear",
                                   "MODEL.DISTILLATOR.TEACHER.SOLVER.LR_SCHEDULER_NAME", "WarmupMultiStepLR", "MODEL.DISTILLATOR.TEACHER.SOLVER.STEPS", "(50,)",
                                   "MODEL.DISTILLATOR.TEACHER.SOLVER.GAMMA", "0.1", "MODEL.DISTILLATOR.TEACHER.SOLVER.WARMUP_FACTOR", "1.0",
                                   "MODEL.DISTILLATOR.TEACHER.SOLVER.WARMUP_ITERS", "0", "MODEL.DISTILLATOR.TEACHER.SOLVER.WARMUP_METHOD", "linear"])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lgd_amd.engine import Trainer
        cfg = _cfg()
        model = _Stub()
        tr = Trainer(cfg, model, device=torch.device("cpu"))
        assert not model.teacher.global_ctx_proj_1D.weight.requires_grad
        data = torch.full((3, 4), float(rank + 1))
        flags, frozen, metrics = [], [], []
        for it in range(6):
            tr.step(data, it)
            flags.append(model.distill_flag)
            frozen.append(not model.student.raw_backbone.weight.requires_grad)
            metrics.append(tr.fetch_metrics()["loss_cls"])
        w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(gathered, w)
        same = all(torch.equal(gathered[0], g) for g in gathered)
        # FCOS packed all-reduce of (num_fg, sum centerness): both ranks must use the rank-AVERAGED counts
        from lgd_amd.student.fcos import FCOSCT
        counts = FCOSCT.reduce_counts(torch.tensor([float(rank + 1), 0.5 * (rank + 1)]))  # rank 0: 1 fg, rank 1: 2 fg
        q.put((rank, flags, frozen, metrics, same, counts.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ddp_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, flags, frozen, metrics, same, lc in res:
        assert flags == [0, 0, 0, 0, 1, 1]            # train.py:184-189 with PRE_NONDISTILL_ITERS=4
        assert frozen == [True, True, False, False, False, False]  # train.py:205-207 with PRE_FREEZE...=2
        assert same                                    # replicas stay identical (gradients were averaged)
    assert res[0][3] == res[1][3]                      # metrics are rank-averaged by the single all-reduce
    assert res[0][5] == [1.5, 0.75] and res[1][5] == [1.5, 0.75]  # one packed all-reduce, mean over ranks


def _real_tree_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lgd_amd import config
        from lgd_amd.distillator import build_model
        from lgd_amd.engine import Trainer
        from torch.nn.parallel import DistributedDataParallel
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        cfg = config.setup_cfg(os.path.join(root, "configs", "lgd_fcos_r50.yaml"),
                               ["MODEL.DEVICE", "cpu", "MODEL.DISTILLATOR.PRE_NONDISTILL_ITERS", "3",
                                "MODEL.DISTILLATOR.PRE_FREEZE_STUDENT_BACKBONE_ITERS", "2", "SOLVER.MAX_ITER", "100"])
        torch.manual_seed(0)
        model = build_model(cfg)  # the REAL DistillatorFCOS module tree: aliased FPN, frozen stem/res2, never-trained ctx proj

        # the HIP kernels have no CPU form, so the forward is replaced by a loss that touches exactly the parameters the real
        # forward trains in each phase (rank-dependent, so that unsynchronised replicas would drift apart)
        def forward(data):
            scale = float(data)
            tot = sum((p.float() ** 2).sum() for p in model.parameters() if p.requires_grad)
            return {"loss_cls": tot * scale * 1e-3, "loss_distill": tot * (1e-4 if model.distill_flag else 0.0)}
        model.forward = forward
        tr = Trainer(cfg, model, device=torch.device("cpu"))
        assert isinstance(tr.model, DistributedDataParallel)
        assert not model.teacher.global_ctx_proj_1D.weight.requires_grad  # no context box: frozen statically
        ddps, frozen = [], []
        for it in range(5):
            tr.step(rank + 1.0, it)
            ddps.append(id(tr.model))
            frozen.append(not model.student.raw_backbone.res5[0].conv1.weight.requires_grad)
        m = tr.fetch_metrics()
        w = torch.cat([p.detach().reshape(-1)[:64] for p in model.parameters()])
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        n_alias = sum(1 for n, _ in model.named_parameters(remove_duplicate=False) if n.startswith("student.fpn."))
        q.put((rank, all(torch.equal(ws[0], x) for x in ws), frozen, len(set(ddps)), m["loss_cls"], n_alias))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_ddp_gloo_world2_real_module_tree():
    """the real Trainer over the real DistillatorFCOS parameter tree under DDP (gloo, world 2): the aliased FPN module
    (`student.backbone` is `student.fpn`), FREEZE_AT parameters, the statically frozen `global_ctx_proj_1D` and the
    phase-frozen backbone all go through reducer construction / rebuild; replicas fed different data stay identical."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_real_tree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, frozen, n_ddp, loss_cls, n_alias in res:
        assert same
        assert frozen == [True, True, False, False, False]
        assert n_ddp == 2          # one reducer per backbone phase
        assert n_alias > 0         # the alias really is registered twice
    assert res[0][4] == res[1][4]  # rank-averaged metric
