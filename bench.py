#!/usr/bin/env python
"""Benchmark of the LGD training step (BASELINE.json metric: images/sec, RetinaNet R-50 + LGD, fwd+bwd).

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: re-executes itself as N ranks, lgd_amd/launch.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W         (the driver's form: runs as the rank it is)

One step = student forward -> dynamic teacher -> head re-run on teacher features -> distillation loss
-> backward -> clip -> both SGD optimizers -> both LR schedulers, on a synthetic COCO-shaped batch
(8 images/GPU of 800x1333 padded to 800x1344, 10 GT boxes each, context box on).
Weak scaling: the per-GPU batch is fixed, `value` = all images of all ranks / max-over-ranks time.

Timing protocol: W warm-up steps, then K steps bracketed by barrier + synchronize give `value` / `ms_per_step` (no per-kernel
instrumentation inside that region); then a SECOND pass of K steps with a HIP event pair around every launch of the library
(recorded on the launch stream) and around the Winograd GEMMs gives the per-kernel numbers.

Extra objects on the JSON line (task statement section 4):
  roofline      -- dominant hand-written HIP kernel of the step (HBM bound): algorithmic bytes per launch / mean launch time
  roofline_mfma -- the Winograd channel GEMMs (library, fp32 MFMA): FLOP per launch / mean launch time vs the 157.3 TF peak
  roofline_mfma_pointwise -- the student's 1x1 convolutions (library GEMMs: forward, input gradient, weight gradient), same form
  host_batch    -- the same steps with the batches handed over as pinned HOST tensors and copied inside every step
  roofline_lgd_forward -- the north star's aggregate "LGD distill forward": gn_pool + box_paint + in_moments, 4 pyramids of bytes
  cpu_baseline  -- the CPU oracle (oracle/lgd_oracle.py, kind "port") timed on the host cores, rank 0, N=1 only, with
                   `gpu_same_path`: the product's SAME sub-path (teacher + adapter + distill loss, fwd+bwd) timed on the GPU
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); exported on the boxes already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _gpus_requested(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__":
    # `python bench.py --gpus N` started plainly: become N ranks of one node (one process per GPU, env rendezvous on 127.0.0.1) BEFORE
    # the runtime is imported; under a launcher (the driver's torch.distributed.run form) this process already is one of the ranks
    from lgd_amd import launch as _launch
    if _gpus_requested(sys.argv[1:]) > 1 and not _launch.launched():
        raise SystemExit(_launch.self_launch(__file__, _gpus_requested(sys.argv[1:]), sys.argv[1:]))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBPS = 6290.0      # what a float4 copy kernel reaches on this part (same guide: "6.29 TB/s measured, 79 %"): reported beside `frac`, never instead of it
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32-input MFMA peak (v_mfma_f32_16x16x4_f32 / 32x32x2_f32), same guide
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16), same guide; the bf16x3 products issue 6 bf16 MFMA flops per fp32 flop


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default=os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"))
    ap.add_argument("--batch-per-gpu", type=int, default=8)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--boxes", type=int, default=10)
    ap.add_argument("--phase", default="distill", choices=["distill", "nondistill", "frozen"],
                    help="training phase to time: distill = steady state (distill on, backbone training; 150k of the "
                         "180k iterations), nondistill = iterations 20k-30k, frozen = first 20k iterations")
    ap.add_argument("--host-batch", action="store_true",
                    help="time ONLY the host-batch protocol (pinned HOST tensors copied inside every step, what the reference's step "
                         "contains: retinanet.py:48) as `value`; default: `value` = batches resident in HBM when the timed region starts "
                         "(task statement section 4) AND the host-batch rate of the same steps in `host_batch` on the same line")
    ap.add_argument("--wino-tile", type=int, default=None, choices=[4, 6], help="F(4x4,3x3) instead of the default F(6x6,3x3) (A/B runs)")
    ap.add_argument("--library-convs", action="store_true", help="every 3x3 convolution on the vendor library instead of winograd*.hip (A/B runs)")
    ap.add_argument("--torch-optimizers", action="store_true", help="torch's multi-tensor clip + SGD instead of the one-launch optim.hip step (A/B runs)")
    ap.add_argument("--no-host-pass", action="store_true", help="skip the extra pass that measures the host-batch rate (profiling runs)")
    ap.add_argument("--batches", type=int, default=4, help="distinct synthetic batches rotated through the steps")
    ap.add_argument("--multiscale", action="store_true",
                    help="per-image short side drawn from INPUT.MIN_SIZE_TRAIN (640..800, max 1333: BASELINE config 5, "
                         "configs/Base-RetinaNet.yaml:26) instead of every image at --height x --width")
    ap.add_argument("--library-gemms", action="store_true", help="the Winograd channel products on the library's fp32 GEMM instead of csrc/gemm3.hip (A/B runs)")
    ap.add_argument("--no-teacher-fold", action="store_true", help="DynamicTeacher: rendering's + ctx / ReLU and the refinement GroupNorm(1) + ReLU pairs as their own passes instead of inside the next convolution's input transform (A/B runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the instrumented second pass")
    ap.add_argument("--one-stream", action="store_true", help="no side streams: the teacher's label encoder, the head's box tower, the adapter and the FPN's small levels run in line on the step's stream (A/B runs; the rocprof summary the per-kernel roofline numbers are checked against)")
    ap.add_argument("--no-gn-bwd-fold", action="store_true", help="FCOS towers: the GroupNorm backward as its own statistics + apply passes instead of inside the producing convolution's adjoint output transform (A/B runs)")
    ap.add_argument("--no-gn-fold", action="store_true", help="FCOS towers: GroupNorm(32) + ReLU as its own passes instead of inside the next convolution's input transform (A/B runs)")
    ap.add_argument("--no-fcos-fused-loss", action="store_true", help="FCOS: GIoU / centerness losses as the composed torch form instead of one kernel on the raw head outputs (A/B runs)")
    ap.add_argument("--head-passes", type=int, default=1, choices=[1, 2],
                    help="1 = one head pass over student + teacher pyramids (default); 2 = the reference's literal two passes")
    ap.add_argument("--cpu-sample-images", type=int, default=1)
    return ap.parse_args()


def cpu_baseline(args, n_boxes, ctx, budget_s=20.0):
    """Oracle LGD path (teacher fwd + adapter + distill loss, fwd+bwd) on the host cores, full-size images.
    Bounded sample: one image is timed first, then as many images (<= the GPU batch) as fit the budget are run as
    one batch.  Threads: the fastest of 16 / 32 / 64 (<= the host's logical CPUs) on the one-image probe, all three reported -- the baseline
    should be the host's best; with one thread per logical CPU (256 on the GPU box) the small per-box ops oversubscribe
    and the same work takes 100x longer (measured), so the sweep stops at 64."""
    from lgd_amd import synth
    from oracle import lgd_oracle as O
    ncpu = os.cpu_count() or 1
    cands = sorted({min(ncpu, n) for n in (16, 32, 64)})
    torch.set_num_threads(cands[0])
    p = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.closed_form_params(O.teacher_param_shapes()).items()}
    pa = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.closed_form_params(O.adapter_param_shapes()).items()}
    H, W = (args.height + 31) // 32 * 32, (args.width + 31) // 32 * 32

    def run(B, h, w):
        feats = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.synth_features(B, h, w, seed=3).items()}
        gt = [(torch.from_numpy(b), torch.from_numpy(c)) for b, c in synth.synth_gt(B, h, w, n_boxes, seed=0)]
        t0 = time.perf_counter()
        tea, _, _ = O.teacher_forward(p, feats, gt, (h, w), add_ctx=ctx)
        loss = O.distill_loss(pa, feats, tea, 1.0, 1) + sum(t.mean() for t in tea.values())
        loss.backward()
        return time.perf_counter() - t0
    run(1, 128, 160)  # one-time library initialisation outside the timed region
    probe = {}
    for n in cands:
        torch.set_num_threads(n)
        run(1, 128, 160)                     # (the thread pool of this size, outside the timed probe)
        probe[n] = min(run(1, H, W), run(1, H, W)) if len(cands) > 1 else run(1, H, W)
    threads = min(probe, key=probe.get)
    torch.set_num_threads(threads)
    t1 = probe[threads]
    B = max(1, min(args.batch_per_gpu, int(budget_s / max(t1, 1e-3))))
    dt, reps = (run(B, H, W) if B > 1 else t1), 1
    while dt < 10.0 and reps < 8:  # aim for >= 10 s of CPU work
        dt += run(B, H, W)
        reps += 1
    B *= reps
    return {"value": B / dt, "unit": "images/sec", "cores": threads, "kind": "port", "gpu_same_path": gpu_lgd_only(args, n_boxes, ctx),
            "threads_probe_s_per_image": {str(n): round(t, 3) for n, t in probe.items()},
            "sample": "oracle LGD path only (dynamic teacher fwd + adapter + distill loss, fwd+bwd; no student "
                      "backbone/head: the reference's detectron2 student is not runnable), %d image(s) of %dx%d, "
                      "%d GT boxes, %.1f s on %d threads (host has %d logical CPUs)"
                      % (B, H, W, n_boxes, dt, threads, os.cpu_count() or 1)}


def gpu_lgd_only(args, n_boxes, ctx, steps=10):
    """the product's DynamicTeacher + adapter + distill loss, forward + backward, on the GPU: the SAME sub-path and inputs the
    CPU baseline times (no student backbone / head), full batch, so that the two numbers can be compared like for like."""
    from lgd_amd import config, ops, synth
    from lgd_amd.adapters import SequentialConvs
    from lgd_amd.dynamic_teacher import DynamicTeacher
    from lgd_amd.structures import Boxes, ImageList, Instances
    from oracle import lgd_oracle as O  # parameter names / shapes only
    dev = torch.device("cuda", torch.cuda.current_device())
    B = args.batch_per_gpu
    H, W = (args.height + 31) // 32 * 32, (args.width + 31) // 32 * 32
    cfg = config.setup_cfg(args.config, ["MODEL.DEVICE", str(dev)])
    teacher = DynamicTeacher(cfg)
    teacher.load_state_dict({k: torch.from_numpy(v) for k, v in synth.closed_form_params(O.teacher_param_shapes()).items()})
    adapter = SequentialConvs(cfg)
    adapter.load_state_dict({k: torch.from_numpy(v) for k, v in synth.closed_form_params(O.adapter_param_shapes()).items()})
    teacher.to(dev).train()
    adapter.to(dev).train()
    feats = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in synth.synth_features(B, H, W, seed=3).items()}
    gt = synth.synth_gt(B, H, W, n_boxes, seed=0)
    bi = [{"image": torch.zeros(3, H, W), "instances": Instances((H, W), gt_boxes=Boxes(torch.from_numpy(b)), gt_classes=torch.from_numpy(c))}
          for b, c in gt]
    images = ImageList(torch.zeros(B, 3, H, W, device=dev), [(H, W)] * B)
    keys = sorted(feats)

    def step():
        tea, _, _ = teacher((bi, images, None, feats))
        stu = adapter.levels([feats[k] for k in keys])
        loss = ops.distill_in_mse(stu, [tea[k] for k in keys], 1.0) + sum(t.mean() for t in tea.values())
        loss.backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": B / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt,
            "sample": "product DynamicTeacher + adapter + distill loss fwd+bwd, %d images of %dx%d, %d GT boxes, %d steps" % (B, H, W, n_boxes, steps)}


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON record: native libraries write banners to the C stdout (RCCL prints its version block at
    # communicator init and libc flushes it at exit, i.e. AFTER the record), so fd 1 is pointed at stderr for the whole run and the
    # record goes to a duplicate of the original stdout
    sys.stdout.flush()
    record_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:   # only reachable when main() is called with a hand-made environment: the __main__ block self-launches
        raise SystemExit("--gpus %d needs %d ranks: run `python bench.py --gpus %d` (self-launching) or torch.distributed.run" % ((args.gpus,) * 3))
    # LGD_BENCH_SHARE_GPU=1 + LGD_BENCH_BACKEND=gloo: every rank on cuda:0, exchange over gloo -- the N > 1 code path of this file
    # (barriers, max over ranks, rank-0 record, DDP) on a 1-GPU box (tests/test_model_gpu.py); never a performance number
    share = os.environ.get("LGD_BENCH_SHARE_GPU", "0") == "1"
    gpu = 0 if share else local_rank
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    force_ddp = os.environ.get("LGD_FORCE_DDP", "0") == "1"  # exercise the RCCL/DDP path on a single GPU (tests)
    pinned_cpus = None
    if world > 1:
        # one slice of the node's CPUs per rank, near its GPU when sysfs tells (lgd_amd/launch.py); LGD_PIN=0 disables
        from lgd_amd import launch
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        try:
            props = [torch.cuda.get_device_properties(0 if share else i) for i in range(local_world)]
            bus = ["%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id) for p in props]
        except Exception:  # noqa: BLE001 -- attribute names differ across torch builds: fall back to equal slices by local rank
            bus = None
        pinned_cpus = launch.pin_host_threads(local_rank, local_world, bus)
    rccl_ranks = None
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("LGD_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        # the communicator really spans `world` ranks: an all-reduce of ones over the process group, read back (also warms the ring up
        # outside the timed region)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(round(float(ones.item())))
        assert rccl_ranks == dist.get_world_size() == max(world, 1), (rccl_ranks, dist.get_world_size(), world)

    from lgd_amd import config, hip, ops
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    hip.load()  # fail loudly if the HIP extension is missing
    if args.wino_tile is not None:
        ops.conv3x3_backend(tile=args.wino_tile)
    if args.library_convs:
        ops.conv3x3_backend(winograd=False)
    if args.library_gemms:
        ops.gemm3_backend(False)

    cfg = config.setup_cfg(args.config, ["MODEL.DEVICE", "cuda:%d" % gpu])
    torch.manual_seed(0)
    model = build_model(cfg)
    model.fused_head_pass = args.head_passes == 1
    if args.no_teacher_fold:
        model.teacher.fold_activations = False
    if args.no_fcos_fused_loss and hasattr(model.student, "fused_reg_loss"):
        model.student.fused_reg_loss = False
    if args.no_gn_fold and hasattr(model.student.head, "fold_group_norm"):
        model.student.head.fold_group_norm = False
    if args.no_gn_bwd_fold and hasattr(model.student.head, "fold_group_norm_bwd"):
        model.student.head.fold_group_norm_bwd = False
    from lgd_amd.student import retinanet as _rn
    from lgd_amd.student import fpn as _fpn
    streams_shipped = (bool(getattr(model.teacher, "side_stream", False)), bool(_rn._HEAD_STREAMS), bool(getattr(model, "adapter_stream", False)), bool(_fpn._FPN_STREAM))

    def side_streams(on):
        """the step's three forks (lgd_amd/streams.py) as shipped, or all off: everything on the step's own stream"""
        model.teacher.side_stream = streams_shipped[0] and on
        _rn._HEAD_STREAMS = streams_shipped[1] and on
        model.adapter_stream = streams_shipped[2] and on
        _fpn._FPN_STREAM = streams_shipped[3] and on
    side_streams(not args.one_stream)
    trainer = Trainer(cfg, model, device=dev, distributed=True if force_ddp else None, fused_sgd=not args.torch_optimizers)
    d = cfg.MODEL.DISTILLATOR
    it0 = {"distill": max(d.PRE_NONDISTILL_ITERS, d.PRE_FREEZE_STUDENT_BACKBONE_ITERS),
           "nondistill": d.PRE_FREEZE_STUDENT_BACKBONE_ITERS, "frozen": 0}[args.phase]
    Bg = args.batch_per_gpu
    from lgd_amd.data import multiscale_sizes
    nb = max(1, args.batches)

    def make_batches(**kw):
        out = []
        for j in range(nb):
            seed = 1000 + rank + 7919 * j
            sizes = (multiscale_sizes(Bg, args.height, args.width, tuple(cfg.INPUT.MIN_SIZE_TRAIN), cfg.INPUT.MAX_SIZE_TRAIN, seed=seed)
                     if args.multiscale else None)
            out.append(synthetic_batch(Bg, args.height, args.width, args.boxes, seed=seed, sizes=sizes, **kw))
        return out
    host_batches = make_batches(pin=True) if (args.host_batch or (world == 1 and not args.no_host_pass)) else None
    batches = host_batches if args.host_batch else make_batches(device=dev)
    ctx = bool(d.TEACHER.ADD_CONTEXT_BOX)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    it = [it0]

    def run_steps(n, data_sets):
        for _ in range(n):
            trainer.step(data_sets[it[0] % nb], it[0])
            it[0] += 1

    def prefetched(n, host_sets):
        """n steps over HOST batches through lgd_amd.data.DevicePrefetcher (the copy of batch k + 1 on a side stream under step k); the
        loader is created -- and its first copy issued -- by the caller, ahead of the timed region, like a loader that runs ahead"""
        from lgd_amd.data import DevicePrefetcher
        return DevicePrefetcher((host_sets[(it[0] + i) % nb] for i in range(n)), dev)

    def run_prefetched(pf):
        for b in pf:
            trainer.step(b, it[0])
            it[0] += 1

    run_steps(max(args.warmup, nb if args.multiscale else 0), batches)   # multi-scale: every batch shape once (library kernel selection)
    trainer.fetch_metrics()
    sync()
    t0 = time.perf_counter()
    run_steps(args.steps, batches)
    sync()
    dt = time.perf_counter() - t0
    metrics = trainer.fetch_metrics()  # raises on non-finite losses
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # the same steps with the batches handed over as pinned host tensors (N = 1 only; never `value` unless --host-batch)
    dt_host = dt_host_inline = None
    if host_batches is not None and not args.host_batch and world == 1 and not args.no_host_pass:
        run_prefetched(prefetched(min(2, args.steps), host_batches))
        pf = prefetched(args.steps, host_batches)
        sync()
        t0 = time.perf_counter()
        run_prefetched(pf)
        sync()
        dt_host = time.perf_counter() - t0
        trainer.fetch_metrics()
        # (and the reference's literal form: the copies inside the step, on the step's stream)
        sync()
        t0 = time.perf_counter()
        run_steps(args.steps, host_batches)
        sync()
        dt_host_inline = time.perf_counter() - t0
        trainer.fetch_metrics()
    # second pass, instrumented: an event pair around every launch of the library and around the Winograd GEMMs
    # The per-kernel numbers are a kernel's OWN durations: the shipped step forks four chains onto side streams, and launches that share the chip
    # stretch each other (the two head towers run their products side by side: each takes ~1.6x as long, the step is 2 % shorter) -- so this pass
    # runs with the forks off, everything on the step's stream, and a third pass as shipped leaves the overlapped durations of the dominant
    # kernel next to them (`roofline.as_shipped`).
    ktimes, kbytes, kflops, dt_instr = {}, {}, {}, None
    ktimes_shipped, dt_instr_shipped = {}, None
    overlapped = any(streams_shipped) and not args.one_stream
    if not args.no_kernel_timing:  # every rank steps (the gradient all-reduce is collective); rank 0 carries the timers
        side_streams(False)
        run_steps(1, batches)
        if rank == 0:
            ops.kernel_timer_enable(True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(args.steps, batches)
        torch.cuda.synchronize()
        dt_instr = time.perf_counter() - t1
        if rank == 0:
            ktimes = ops.kernel_timer_collect()
            kbytes = ops.kernel_alg_bytes()   # shape-varying kernels (Winograd transforms): bytes summed over the timed launches
            kflops = ops.kernel_gemm_flops()
            ops.kernel_timer_enable(False)
        side_streams(not args.one_stream)
        if overlapped:
            run_steps(1, batches)
            if rank == 0:
                ops.kernel_timer_enable(True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(args.steps, batches)
            torch.cuda.synchronize()
            dt_instr_shipped = time.perf_counter() - t1
            if rank == 0:
                ktimes_shipped = ops.kernel_timer_collect()
                ops.kernel_timer_enable(False)
    if world > 1:
        dist.barrier()

    if rank == 0:
        Hp, Wp = (args.height + 31) // 32 * 32, (args.width + 31) // 32 * 32
        from lgd_amd import synth
        # one fp32 pyramid, bytes: the padded batch tensor's (multi-scale: the mean over the rotated batches, each padded to its own maximum)
        pads = [((max(x["height"] for x in b) + 31) // 32 * 32, (max(x["width"] for x in b) + 31) // 32 * 32) for b in batches]
        px_pyr = sum(sum(h * w for h, w in synth.pyramid_shapes(hp, wp)) for hp, wp in pads) / len(pads)
        P = Bg * 256 * px_pyr * 4
        # algorithmic bytes per launch
        alg = {"in_moments_kernel": 2 * P, "in_mse_bwd_kernel": 3 * P, "box_sum_kernel": P, "box_paint_kernel": P,
               "gn_stats_kernel": P, "gn_apply_kernel": 2 * P, "gn_bwd_stats_kernel": 2 * P, "gn_bwd_apply_kernel": 3 * P,
               "ctx_relu_kernel": 2 * P, "ctx_relu_bwd_kernel": 3 * P,
               "gn_pool_kernel": P, "gn_pool_bwd_apply_kernel": 2 * P}
        # focal loss: logits (N, 9*80, H, W) read once (fwd) / read + written (bwd); int32 label planes (N, 9, H, W) on top
        Pf = Bg * px_pyr * 9 * 4
        alg.update({"focal_fwd_kernel": 80 * Pf + Pf, "focal_bwd_kernel": 2 * 80 * Pf + Pf, "focal_fwd_grad_kernel": 2 * 80 * Pf + Pf})
        alg.update({k: v / max(ktimes[k][0], 1) for k, v in kbytes.items() if k in ktimes})  # mean bytes per launch
        kernels = {}
        for name, (n, ms, lo, hi) in ktimes.items():
            kernels[name] = {"launches": n, "avg_us": 1e3 * ms / max(n, 1), "min_us": 1e3 * lo, "max_us": 1e3 * hi, "total_ms": ms}
            if name in alg:
                kernels[name]["alg_bytes"] = alg[name]
                kernels[name]["GBps"] = alg[name] / (1e-3 * ms / n) / 1e9
            if name in kflops:
                kernels[name]["TFLOPs"] = kflops[name] / (1e-3 * ms) / 1e12
        is_cfg1 = (os.path.abspath(args.config) == os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml") and Bg == 8
                   and (args.height, args.width) == (800, 1333))
        hbm_bound = [k for k in kernels if k in alg and not k.startswith("focal") and k != "gemm3_kernel" and "amax" not in k]  # focal is exp/log bound, gemm3 MFMA bound
        dom = max(hbm_bound, key=lambda k: kernels[k]["total_ms"], default=None)
        roofline = roofline_mfma = None
        PMC = "r06_pmc_traffic.json"

        def pmc_traffic(name):
            """HBM bytes per launch of `name` from the tracked counter summary (tools/pmc_bench.sh collects the CSV and this JSON in ONE command)
            -- refused when its launches per step are not this run's: a stale file must not price another launch mix"""
            tf = os.path.join(ROOT, "profiles", PMC)
            if not (os.path.exists(tf) and is_cfg1 and name in kernels):   # the counters were collected on configs[1]
                return None, None
            e = json.load(open(tf)).get(name)
            if not e or not e.get("steps_profiled"):
                return None, None
            mine, theirs = kernels[name]["launches"] / args.steps, e["launches"] / e["steps_profiled"]
            if abs(mine - theirs) > 0.01:
                return None, "profiles/%s REFUSED: %.2f launches of %s per step there, %.2f in this run -- re-run tools/pmc_bench.sh" % (PMC, theirs, name, mine)
            return e["hbm_bytes_per_launch"], ("static profile: profiles/%s (rocprofv3 --pmc FETCH_SIZE x 2 / WRITE_SIZE passes over this command, "
                                               "%.0f launches per step as in this run; not re-measured in this run)" % (PMC, theirs))
        if dom:
            traffic, tsrc = pmc_traffic(dom)
            k = kernels[dom]
            roofline = {"bound": "hbm", "kernel": dom, "achieved": k["GBps"], "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": k["GBps"] / HBM_PEAK_GBPS, "frac_of_measured_copy_rate": k["GBps"] / HBM_COPY_GBPS, "traffic": traffic,
                        "traffic_source": tsrc,
                        "alg_bytes_per_launch": alg[dom], "avg_launch_us": k["avg_us"], "min_launch_us": k["min_us"],
                        "max_launch_us": k["max_us"], "ms_per_step": k["total_ms"] / args.steps,
                        # (the f16x2 products: MFMA work issued = 3 f16 MFMA flops per fp32 flop, against the 2.5 PFLOP/s dense f16 peak -- they are HBM-bound)
                        "mfma_frac_of_f16_peak": (3.0 * kflops[dom] / (1e-3 * k["total_ms"]) / 1e12 / MFMA_BF16_PEAK_TFLOPS) if dom in kflops else None,
                        "fp32_equivalent_TFLOPs": (kflops[dom] / (1e-3 * k["total_ms"]) / 1e12) if dom in kflops else None,
                        "all_hip_kernels": {n: {"avg_us": round(v["avg_us"], 2), "min_us": round(v["min_us"], 2),
                                                "max_us": round(v["max_us"], 2), "GBps": round(v.get("GBps", 0.0), 1),
                                                "launches_per_step": v["launches"] / args.steps}
                                            for n, v in kernels.items() if "_gemm" not in n}}
        # the dominant hand-written kernel of the step overall: since round 4 that is gemm3_kernel (MFMA bound), not a transform.  `roofline`
        # names whichever has more in-step time; the dominant HBM-bound kernel stays on the line as `roofline_hbm`
        roofline_hbm = roofline
        # the dominant hand-written kernel of the step overall may be an MFMA-bound product kernel (csrc/gemm3.hip: bf16x3, 6 MFMA flops per fp32 flop;
        # its f16x2 form lgd_gemm2h: 3) instead of an HBM-bound one: `roofline` names whichever has most in-step time; the dominant HBM-bound
        # kernel stays on the line as `roofline_hbm`
        for kname, tag, mult, what in (("gemm3_kernel", "_gemm3_", 6.0, "bf16 MFMA flops issued = 6 per fp32 multiply-add pair (three-way split operands, 6 of 9 cross products)"),
                                       ("gemm2h_kernel", "_gemm2h_", 3.0, "f16 MFMA flops issued = 3 per fp32 multiply-add pair (two-piece split operands, 3 of 4 cross products)")):
            fl = sum(v for n, v in kflops.items() if tag in n)
            best = kernels[roofline["kernel"]]["total_ms"] if roofline else 0.0
            if kname in kernels and fl and kernels[kname]["total_ms"] > best:
                k = kernels[kname]
                ach = mult * fl / (1e-3 * k["total_ms"]) / 1e12
                traffic, tsrc = pmc_traffic(kname)
                roofline = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                            "flop_per_launch": mult * fl / k["launches"], "fp32_equivalent_TFLOPs": fl / (1e-3 * k["total_ms"]) / 1e12,
                            "alg_bytes_per_launch": kbytes.get(kname, 0) / max(k["launches"], 1) or None,
                            "avg_launch_us": k["avg_us"], "min_launch_us": k["min_us"], "max_launch_us": k["max_us"],
                            "ms_per_step": k["total_ms"] / args.steps,
                            "note": what + "; the kernel's launches alone (HIP events inside the library), filter-image launches not included",
                            "all_hip_kernels": (roofline_hbm or {}).get("all_hip_kernels")}
        if roofline is not roofline_hbm and roofline_hbm:
            roofline_hbm = {k_: v for k_, v in roofline_hbm.items() if k_ != "all_hip_kernels"}
        # the north star's aggregate: the LGD distill forward = mask pooling (fused with GN + ReLU) + rendering paint + distill moments,
        # one launch each per step, 4 P of algorithmic bytes together
        lgd_fwd = None
        names = ("gn_pool_kernel", "box_paint_kernel", "in_moments_kernel")
        if all(n in kernels for n in names):
            us = sum(kernels[n]["avg_us"] for n in names)
            lgd_fwd = {"bound": "hbm", "kernels": list(names), "alg_bytes": 4 * P, "us": us, "achieved": 4 * P / us / 1e3, "peak": HBM_PEAK_GBPS,
                       "unit": "GB/s", "frac": 4 * P / us / 1e3 / HBM_PEAK_GBPS}
        roofline_pw = None
        pw = {n: v for n, v in kernels.items() if n.startswith("pw_gemm_")}
        if pw:
            tot_ms = sum(v["total_ms"] for v in pw.values())
            tot_fl = sum(kflops[n] for n in pw)
            n_l = sum(v["launches"] for v in pw.values())
            ach = tot_fl / (1e-3 * tot_ms) / 1e12
            roofline_pw = {"bound": "mfma", "kernel": "the student's 1x1 convolutions (library GEMMs: forward = MIOpen's GEMM path, input gradient, "
                           "weight gradient as per-image NT GEMMs)", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None, "flop_per_launch": tot_fl / n_l,
                           "avg_launch_us": 1e3 * tot_ms / n_l, "ms_per_step": tot_ms / args.steps,
                           "by_kind": {n: {"TFLOPs": round(v["TFLOPs"], 1), "avg_us": round(v["avg_us"], 1), "min_us": round(v["min_us"], 1),
                                           "max_us": round(v["max_us"], 1), "launches_per_step": v["launches"] / args.steps}
                                       for n, v in pw.items()}}
        def mfma_object(sel, label, peak, mult):
            tot_ms = sum(v["total_ms"] for v in sel.values())
            tot_fl = sum(kflops[n] for n in sel)
            n_l = sum(v["launches"] for v in sel.values())
            ach = mult * tot_fl / (1e-3 * tot_ms) / 1e12
            return {"bound": "mfma", "kernel": label, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                    "flop_per_launch": mult * tot_fl / n_l, "fp32_equivalent_TFLOPs": tot_fl / (1e-3 * tot_ms) / 1e12,
                    "avg_launch_us": 1e3 * tot_ms / n_l, "ms_per_step": tot_ms / args.steps,
                    "by_kind": {n: {"TFLOPs": round(mult * v["TFLOPs"], 1), "avg_us": round(v["avg_us"], 1), "min_us": round(v["min_us"], 1),
                                    "max_us": round(v["max_us"], 1), "launches_per_step": v["launches"] / args.steps} for n, v in sel.items()}}
        # the Winograd channel products: csrc/gemm3.hip (forward, input gradient; bf16 MFMA, 6 MFMA flops per fp32 flop, split + product launch
        # timed together) and the library's fp32 GEMMs (weight gradient; shapes outside gemm3's tile)
        g3 = {n: v for n, v in kernels.items() if n.startswith("wino_gemm3_") or n.startswith("pw_gemm3_")}
        gl = {n: v for n, v in kernels.items() if n.startswith("wino_gemm_")}
        g2 = {n: v for n, v in kernels.items() if n.startswith("pw_gemm2h_")}
        roofline_mfma_1x1 = (mfma_object(g2, "lgd_gemm2h (csrc/gemm3.hip, f16x2 form): the student's 1x1 convolutions, forward + input gradient (filter as a two-piece f16 image, "
                                         "activations split in registers) and weight gradient (lgd_h2_pwdw: both operands split in registers); 3 of 4 cross products on "
                                         "v_mfma_f32_32x32x16_f16, fp32 accumulate; image launch included",
                                         MFMA_BF16_PEAK_TFLOPS, 3.0) if g2 else None)
        roofline_mfma_lib = None
        if g3:
            roofline_mfma = mfma_object(g3, "csrc/gemm3.hip: Winograd channel products and the student's 1x1 convolutions, forward + input gradient (fp32 operands "
                                        "split into three bf16 pieces, 6 of 9 cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulate; filter split launch included)",
                                        MFMA_BF16_PEAK_TFLOPS, 6.0)
        if gl:
            roofline_mfma_lib = mfma_object(gl, "Winograd channel GEMMs left on the library (fp32 MFMA via torch.bmm: weight gradient, shapes outside gemm3's tile)",
                                            MFMA_F32_PEAK_TFLOPS, 1.0)
            if not g3:
                roofline_mfma, roofline_mfma_lib = roofline_mfma_lib, None
        def h2_object(name, what):
            if name not in kernels or name not in alg:
                return None
            k = kernels[name]
            traffic, tsrc = pmc_traffic(name)
            return {"bound": "hbm", "kernel": name, "what": what, "achieved": k["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": k["GBps"] / HBM_PEAK_GBPS,
                    "traffic": traffic, "traffic_source": tsrc, "alg_bytes_per_launch": alg[name], "avg_launch_us": k["avg_us"], "min_launch_us": k["min_us"],
                    "max_launch_us": k["max_us"], "launches_per_step": k["launches"] / args.steps, "ms_per_step": k["total_ms"] / args.steps,
                    "fp32_equivalent_TFLOPs": kflops[name] / (1e-3 * k["total_ms"]) / 1e12,
                    "mfma_frac_of_f16_peak": 3.0 * kflops[name] / (1e-3 * k["total_ms"]) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
        roofline_h2 = {"forward_and_input_gradient": h2_object("h2_fwd_kernel", "M = U V and dV = U^T dM of the F(6x6,3x3) convolutions: filter image x f16x2 split rows"),
                       "weight_gradient": h2_object("h2_dw_kernel", "dU = dM V^T: split rows x split rows, split-K (partials added by the filter transform's adjoint)")}
        arch = cfg.MODEL.META_ARCHITECTURE.replace("Distillator", "")
        default_cfg = os.path.abspath(args.config) == os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml")
        out = {
            "metric": "images/sec/node RetinaNet R-50+LGD fwd+bwd" if default_cfg else
                      "images/sec/node %s R-%d+LGD fwd+bwd" % (arch, cfg.MODEL.RESNETS.DEPTH),
            "value": world * Bg * args.steps / dt,
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "rccl_ranks": rccl_ranks,   # sum of ones over the process group after init (None: no process group at N = 1)
            "collective_backend": (dist.get_backend() if (world > 1 or force_ddp) else None),
            "host_threads_pinned": None if pinned_cpus is None else "%d CPUs per rank (rank 0: %d-%d)" % (len(pinned_cpus), pinned_cpus[0], pinned_cpus[-1]),
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (f16x2 split products, fp32 accumulate)" if (ops._H2_ON and ops._GEMM3_ON and ops._GEMM2H_ON) else
                      "f32 (3x3 convolutions: f16x2 split products, 1x1: bf16x3 split products; fp32 accumulate)" if (ops._H2_ON and ops._GEMM3_ON) else
                      "f32 (bf16x3 split products, fp32 accumulate)" if ops._GEMM3_ON else "f32"),
            "winograd_products": "csrc/h2.hip (f16x2 operands split in HBM by the transforms: forward, input and weight gradient)" if ops._H2_ON else
                                 ("csrc/gemm3.hip (forward, input gradient) + library (weight gradient)" if ops._GEMM3_ON else "library"),
            "filter_images_from_transform": bool(ops._GEMM3_ON and ops._FILTER_IMAGES),
            "data": ("synthetic (%d distinct batches rotated; %s)" % (nb, "pinned HOST batches copied inside every step" if args.host_batch
                                                                      else "device-resident when the timed region starts, no H2D copy inside it")),
            "host_batch": None if dt_host is None else {
                "value": Bg * args.steps / dt_host, "unit": "images/sec", "ms_per_step": 1e3 * dt_host / args.steps,
                "copies_inside_the_step": {"value": Bg * args.steps / dt_host_inline, "ms_per_step": 1e3 * dt_host_inline / args.steps},
                "note": "the same %d steps with the batches handed over as pinned HOST tensors: the reference's step contains the copy "
                        "(retinanet.py:48); here lgd_amd.data.DevicePrefetcher issues the copy of batch k + 1 on a side stream under step k "
                        "(`value`), next to the literal form with the copies inside the step on its own stream (`copies_inside_the_step`)" % args.steps},
            "config": {"workload": "%s%s R-%d FPN + LGD, %d img/GPU, %dx%d (padded %dx%d), "
                                   "%d GT boxes/img, ctx box %s, phase=%s (fwd+bwd+clip+2xSGD)%s"
                                   % ("BASELINE configs[1]: " if default_cfg and Bg == 8 and not args.multiscale else "", arch,
                                      cfg.MODEL.RESNETS.DEPTH, Bg, args.height, args.width, Hp, Wp, args.boxes, "on" if ctx else "off", args.phase,
                                      ", multi-scale: short side per image from %s" % (list(cfg.INPUT.MIN_SIZE_TRAIN),) if args.multiscale else ""),
                       "yaml": os.path.relpath(args.config, ROOT), "global_batch": world * Bg, "parallelism": "dp%d" % world},
            "losses": {k: round(v, 6) for k, v in metrics.items()},
            "losses_parity": {"loss_distill": "pinned: reference golden fixtures (tests/golden, generated from the reference), teacher features <= 1e-4",
                              "detection_losses": ("FCOS head + target assignment pinned to the reference's in-tree code; focal / GIoU: restated (cvpods absent)"
                                                   if arch == "FCOS" else
                                                   "unpinned (detectron2 0.3 / fvcore absent from the reference tree and the image): "
                                                   "kernels are held to oracle/student_oracle.py, a restatement of the public definitions")},
            "gemm_solution_table_loaded": bool(trainer.tuned_gemms), "conv3x3": "winograd F(%dx%d,3x3)" % (ops._WINO_TILE, ops._WINO_TILE) if ops._WINO_ON else "library",
            "hbm_peak_alloc_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
            "ms_per_step_instrumented": None if dt_instr is None else 1e3 * dt_instr / args.steps,
            "side_streams": {"teacher_label_encoder": streams_shipped[0] and not args.one_stream, "head_class_tower": streams_shipped[1] and not args.one_stream,
                             "adapter": streams_shipped[2] and not args.one_stream, "fpn_small_levels": streams_shipped[3] and not args.one_stream,
                             "note": "`value` / `ms_per_step` are the step as shipped (the forks on); the per-kernel durations behind every roofline object are taken in a pass "
                                     "with the forks OFF (one stream: a kernel's own duration -- launches that share the chip stretch each other); `roofline.as_shipped` "
                                     "carries the dominant kernel's overlapped durations from a third pass; --one-stream runs everything without the forks"},
            "ms_per_step_instrumented_as_shipped": None if dt_instr_shipped is None else 1e3 * dt_instr_shipped / args.steps,
            "fused_clip_sgd": trainer._fused_sgd is not None,
            "head_pass": "single (student + teacher pyramids in one pass)" if getattr(model, "fused_head_pass", False) else "two passes",
            "roofline": roofline, "roofline_hbm": roofline_hbm if roofline_hbm is not roofline else None, "roofline_mfma": roofline_mfma, "roofline_mfma_library": roofline_mfma_lib, "roofline_mfma_pointwise": roofline_pw, "roofline_mfma_1x1_f16x2": roofline_mfma_1x1,
            "roofline_lgd_forward": lgd_fwd, "roofline_h2_products": roofline_h2 if any(roofline_h2.values()) else None,
        }
        if roofline and ktimes_shipped.get(roofline.get("kernel")):
            n_, ms_, lo_, hi_ = ktimes_shipped[roofline["kernel"]]
            ab = roofline.get("alg_bytes_per_launch")
            roofline["as_shipped"] = {"avg_launch_us": 1e3 * ms_ / max(n_, 1), "min_launch_us": 1e3 * lo_, "max_launch_us": 1e3 * hi_, "ms_per_step": ms_ / args.steps,
                                      "achieved": (ab / (1e-3 * ms_ / max(n_, 1)) / 1e9) if (ab and roofline.get("bound") == "hbm") else None,
                                      "note": "the same launches with the step's side streams on: two chains run their launches side by side, each launch shares the chip"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.boxes, ctx)
        record_out.write(json.dumps(out) + "\n")
        record_out.flush()
    if world > 1 or force_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
