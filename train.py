#!/usr/bin/env python
"""Training entry with the reference's command line
    python train.py --config-file CFG --num-gpus N [--resume] [--eval-only] [KEY VALUE ...]
[ref: train.py:237-310; README.md:97-126].  N > 1: the command above re-executes itself as one process per GPU (lgd_amd/launch.py), or
start the ranks yourself with
    python -m torch.distributed.run --nproc-per-node N train.py --config-file CFG --num-gpus N ...
(RCCL over xGMI; the reference's detectron2 `launch` + tcp rendezvous is replaced by the env rendezvous).
Data: synthetic COCO-shaped batches (lgd_amd/data.py) -- real-image IO/evaluation are out of scope.
Checkpoints carry the reference's keys {model, stu_optimizer, tea_optimizer, stu_scheduler, tea_scheduler,
iteration} [ref: train.py:155-167]; metrics go to OUTPUT_DIR/metrics.json every 20 iterations with the
reference's scalar names [ref: train.py:199,212-213,229-233]."""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _num_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--num-gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--num-gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__":
    # the reference's `launch(main, args.num_gpus, ...)` (train.py:303-310): started plainly with --num-gpus N, become N ranks of this node
    from lgd_amd import launch as _launch
    if _num_gpus(sys.argv[1:]) > 1 and not _launch.launched():
        raise SystemExit(_launch.self_launch(__file__, _num_gpus(sys.argv[1:]), sys.argv[1:]))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-file", default=os.path.join(ROOT, "configs", "lgd_retinanet_r50.yaml"))
    ap.add_argument("--num-gpus", type=int, default=1)
    ap.add_argument("--resume", action="store_true")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--max-iter", type=int, default=None, help="stop early (synthetic runs)")
    ap.add_argument("--image-size", type=int, nargs=2, default=(800, 1333))
    ap.add_argument("opts", nargs=argparse.REMAINDER, default=[])
    return ap.parse_args()


def latest_checkpoint(out_dir):
    f = os.path.join(out_dir, "last_checkpoint")
    if os.path.exists(f):
        return os.path.join(out_dir, open(f).read().strip())
    return None


def save_checkpoint(trainer, out_dir, name):
    os.makedirs(out_dir, exist_ok=True)
    torch.save(trainer.state_dict(), os.path.join(out_dir, name))
    with open(os.path.join(out_dir, "last_checkpoint"), "w") as f:
        f.write(name)


def main():
    args = parse()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if args.num_gpus > 1 and world != args.num_gpus:
        raise SystemExit("--num-gpus %d but WORLD_SIZE=%d" % (args.num_gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        from lgd_amd import launch
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        try:
            props = [torch.cuda.get_device_properties(i) for i in range(local_world)]
            bus = ["%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id) for p in props]
        except Exception:  # noqa: BLE001
            bus = None
        launch.pin_host_threads(local_rank, local_world, bus)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from lgd_amd import config
    from lgd_amd.data import synthetic_batch
    from lgd_amd.distillator import build_model
    from lgd_amd.engine import Trainer
    cfg = config.setup_cfg(args.config_file, ["MODEL.DEVICE", "cuda:%d" % local_rank] + [o for o in args.opts if o != "--"])
    torch.manual_seed(0 if cfg.SEED < 0 else cfg.SEED)
    model = build_model(cfg)
    model.distill_flag = cfg.MODEL.DISTILLATOR.DISTILL_OFF  # train.py:266
    per_gpu = max(1, cfg.SOLVER.IMS_PER_BATCH // world)
    h, w = args.image_size
    from lgd_amd.checkpoint import load_checkpoint
    weights = cfg.MODEL.WEIGHTS if cfg.MODEL.WEIGHTS and os.path.isfile(cfg.MODEL.WEIGHTS) else None
    if cfg.MODEL.WEIGHTS and weights is None and rank == 0:
        print("MODEL.WEIGHTS %r is not a local file: starting from the constructed initialisation" % cfg.MODEL.WEIGHTS)
    if args.eval_only:
        ck = (latest_checkpoint(cfg.OUTPUT_DIR) if args.resume else None) or weights  # resume_or_load, train.py:270-272
        if ck:
            rep = load_checkpoint(ck, model)
            if rank == 0:
                print("loaded %s: %s" % (ck, rep))
        model.eval()
        with torch.no_grad():
            for probe in ([True, False] if cfg.MODEL.DISTILLATOR.EVAL_TEACHER else [False]):  # train.py:268-276
                out = model(synthetic_batch(per_gpu, h, w, 10, seed=7), eval_teacher=probe)
                if rank == 0:
                    print("eval_teacher=%s: %d images, %d detections" % (probe, len(out), sum(len(o["instances"]) for o in out)))
        return
    trainer = Trainer(cfg, model, device=dev)
    start = 0
    ck = latest_checkpoint(cfg.OUTPUT_DIR) if args.resume else None
    if ck:  # [ref: train.py:159-161] resume: weights + both optimizers + both schedulers + iteration
        rep = load_checkpoint(ck, model, trainer, resume=True)
        start = trainer.iteration
    elif weights:  # fresh run from MODEL.WEIGHTS (ImageNet backbone pickle or a released LGD checkpoint): weights only
        rep = load_checkpoint(weights, model)
    if (ck or weights) and rank == 0:
        print("loaded %s: %s" % (ck or weights, rep))
        if rep.unexpected:
            print("  unexpected keys:", rep.unexpected[:8], "..." if len(rep.unexpected) > 8 else "")
    max_iter = cfg.SOLVER.MAX_ITER if args.max_iter is None else min(cfg.SOLVER.MAX_ITER, args.max_iter)
    metrics_f = open(os.path.join(cfg.OUTPUT_DIR, "metrics.json"), "a") if rank == 0 and (os.makedirs(cfg.OUTPUT_DIR, exist_ok=True) or True) else None
    t0 = time.perf_counter()
    # the synthetic loader (0.5 s of numpy per 16 images) runs one batch ahead on a worker thread, pinned, like a DataLoader would
    from concurrent.futures import ThreadPoolExecutor
    loader = ThreadPoolExecutor(1)
    make = lambda i: synthetic_batch(per_gpu, h, w, 10, seed=i * world + rank, pin=True)  # noqa: E731
    nxt = loader.submit(make, start) if start < max_iter else None
    for it in range(start, max_iter):
        data = nxt.result()
        nxt = loader.submit(make, it + 1) if it + 1 < max_iter else None
        trainer.step(data, it)
        if (it + 1) % 20 == 0 or it == max_iter - 1:  # train.py:229-233
            m = trainer.fetch_metrics()  # one all-reduce + one host copy per log period; raises on non-finite loss
            if rank == 0:
                m.update(iteration=it, time=(time.perf_counter() - t0) / (it + 1 - start))
                metrics_f.write(json.dumps(m) + "\n")
                metrics_f.flush()
                print("iter %d  total_loss %.4f  %s  stu_lr %.6f  %.3f s/iter" % (
                    it, m["total_loss"], "  ".join("%s %.4f" % (k, v) for k, v in m.items() if k.startswith("loss")), m["stu_lr"], m["time"]))
        if (it + 1) % cfg.SOLVER.CHECKPOINT_PERIOD == 0 or it == max_iter - 1:  # train.py:165-167,234
            trainer.check_finite()  # never write (and point last_checkpoint at) weights that went through a non-finite loss
            if rank == 0:
                save_checkpoint(trainer, cfg.OUTPUT_DIR, "model_%07d.pth" % it if it != max_iter - 1 else "model_final.pth")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
