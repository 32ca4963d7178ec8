"""Host-side placement of one rank of a multi-GPU run: CPU affinity next to its GPU, thread caps.

One process per GPU (BASELINE config 5).  Each rank drives its GPU from a handful of host threads --
the Python thread inside the C ABI, the horizon-group pool of ``IlqrCandidateEvaluator``, the out-of-process
``hipcc`` builds of shape plugins (``csrc/jit_host.hpp``; child processes inherit the affinity), numpy / torch
CPU work between launches.  Eight unpinned ranks on a two-socket host migrate between sockets and share every
core's caches; what a rank needs is the cores of ITS GPU's NUMA node, divided among the ranks whose GPUs hang
off the same node, and pools no larger than that share.

``pin_rank`` reads the GPU's NUMA node from sysfs (``/sys/bus/pci/devices/<bdf>/numa_node`` with the PCI address
torch reports), takes this rank's slice of the node's CPUs (``/sys/devices/system/node/nodeN/cpulist``), applies it
with ``os.sched_setaffinity`` and caps the OpenMP / MKL / torch thread pools at the slice.  Where sysfs has no
answer (containers, single-node hosts) the allowed CPUs are split evenly by local rank.  ``AMPC_PIN=0`` disables it.
"""
import os


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(device_index):
    """NUMA node of a visible GPU, or None when the platform does not say."""
    try:
        import torch
        prop = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(prop, "pci_domain_id", 0), prop.pci_bus_id, prop.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        return node if node >= 0 else None
    except Exception:
        return None


def node_cpus(node):
    try:
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            return _parse_cpulist(f.read())
    except Exception:
        return None


def rank_cpus(local_rank, local_world, allowed, nodes=None):
    """This rank's CPUs out of `allowed`: with `nodes` ({rank: (node, cpus of the node)} for every local rank) the
    rank's node is divided among the ranks that share it, in rank order; without, `allowed` is divided evenly.
    Every rank gets at least one CPU; deterministic, so ranks agree without talking."""
    allowed = sorted(allowed)
    if nodes and nodes.get(local_rank) and nodes[local_rank][1]:
        node, cpus = nodes[local_rank]
        pool = [c for c in cpus if c in set(allowed)] or allowed
        mates = sorted(r for r, v in nodes.items() if v and v[0] == node)
    else:
        pool, mates = allowed, list(range(local_world))
    k, n = mates.index(local_rank) if local_rank in mates else 0, max(len(mates), 1)
    share = max(1, len(pool) // n)
    mine = pool[k * share:(k + 1) * share] if k < n - 1 else pool[k * share:]
    return mine or [pool[k % len(pool)]]


def pin_rank(local_rank, local_world, set_thread_caps=True):
    """Pin the calling process for local rank `local_rank` of `local_world`; returns a record of what was done
    ({"cpus": [...], "numa_node": n or None, "threads": cap}) or None when pinning is off / unavailable."""
    if os.environ.get("AMPC_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    nodes = {}
    for r in range(local_world):
        node = gpu_numa_node(r)
        nodes[r] = (node, node_cpus(node)) if node is not None else None
    mine = rank_cpus(local_rank, local_world, allowed, nodes)
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    cap = len(mine)
    if set_thread_caps:
        for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
            os.environ[var] = str(cap)
        try:
            import torch
            torch.set_num_threads(cap)
        except Exception:
            pass
    return {"cpus": mine, "numa_node": nodes[local_rank][0] if nodes.get(local_rank) else None, "threads": cap}


def thread_cap(default):
    """A pool size no larger than the CPUs this process may run on."""
    try:
        return max(1, min(int(default), len(os.sched_getaffinity(0))))
    except AttributeError:
        return int(default)
