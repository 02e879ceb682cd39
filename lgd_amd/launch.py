"""One process per GPU on ONE node: self-launch and host-thread placement  [ref: train.py:296-310 -- detectron2 `launch(main,
num_gpus, ...)` spawns `num_gpus` workers and rendezvous over tcp://127.0.0.1:port].

`python bench.py --gpus N` / `python train.py --num-gpus N` started WITHOUT a launcher re-execute themselves under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (env rendezvous,
backend nccl = RCCL over xGMI); started under a launcher (WORLD_SIZE in the environment) they run as the rank they are.
This module imports no torch: the parent of a self-launch never initialises the runtime.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launched():
    """True when this process is a rank of a torch.distributed launch (env rendezvous variables present)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def self_launch(script, n, argv):
    """re-execute `script argv...` as `n` ranks of one node; returns the launcher's exit code.  stdout / stderr are inherited, so
    rank 0's ONE JSON line is the only thing on stdout (the launcher itself logs to stderr)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(script)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL fails with hipIpcGetMemHandle otherwise on this pool
    # the launcher exports OMP_NUM_THREADS=1 when it is unset; each rank sizes its own pool in pin_host_threads()
    return subprocess.run(cmd, env=env).returncode


def _parse_cpulist(s):
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _gpu_numa_node(pci_bus_id):
    """NUMA node of a GPU from sysfs (`0000:c1:00.0` -> /sys/bus/pci/devices/.../numa_node); None when unknown."""
    try:
        with open("/sys/bus/pci/devices/%s/numa_node" % pci_bus_id.lower()) as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except (OSError, ValueError):
        return None


def plan_affinity(local_rank, local_world, allowed, node_of_rank=None, cpus_of_node=None):
    """CPUs for one rank.  With topology (`node_of_rank[r]` = NUMA node of rank r's GPU, `cpus_of_node[n]` = that node's CPUs): the
    ranks whose GPUs hang off one node share that node's allowed CPUs in contiguous equal slices; without: contiguous equal slices
    of the allowed set by local rank.  Never returns an empty set (falls back to every allowed CPU)."""
    allowed = sorted(allowed)
    if node_of_rank and cpus_of_node and node_of_rank.get(local_rank) is not None:
        node = node_of_rank[local_rank]
        mates = sorted(r for r, n in node_of_rank.items() if n == node)
        cpus = [c for c in cpus_of_node.get(node, []) if c in set(allowed)]
        k = len(cpus) // max(len(mates), 1)
        if k > 0:
            i = mates.index(local_rank)
            return cpus[i * k:(i + 1) * k]
    k = len(allowed) // max(local_world, 1)
    if k == 0:
        return allowed
    return allowed[local_rank * k:(local_rank + 1) * k]


def pin_host_threads(local_rank, local_world, pci_bus_ids=None, max_threads=8):
    """bind this rank's host threads to its own slice of the node's CPUs (near its GPU when sysfs tells) and size the intra-op pool:
    at 2 images per GPU the host issues ~25 ms of Python + dispatch per ~30 ms step (profiles/r03_cpu_issue_time_config4.txt), and
    eight ranks that float over two sockets share caches and steal each other's cores.  LGD_PIN=0 disables.  Returns the CPU list."""
    if os.environ.get("LGD_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = os.sched_getaffinity(0)
    node_of_rank, cpus_of_node = None, None
    if pci_bus_ids:
        node_of_rank = {r: _gpu_numa_node(b) for r, b in enumerate(pci_bus_ids)}
        cpus_of_node = {}
        for n in set(v for v in node_of_rank.values() if v is not None):
            try:
                with open("/sys/devices/system/node/node%d/cpulist" % n) as f:
                    cpus_of_node[n] = _parse_cpulist(f.read())
            except OSError:
                pass
    cpus = plan_affinity(local_rank, local_world, allowed, node_of_rank, cpus_of_node)
    if not cpus:
        return None
    # every thread of the process, not only the calling one: the HIP runtime's worker / signal threads exist as soon as the device has been
    # touched and keep their old mask otherwise; threads created later inherit the caller's
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
        except OSError:
            pass   # (a thread that exited in between)
    os.sched_setaffinity(0, cpus)
    n = max(1, min(len(cpus), max_threads))
    os.environ["OMP_NUM_THREADS"] = str(n)
    if "torch" in sys.modules:
        sys.modules["torch"].set_num_threads(n)
    return cpus
