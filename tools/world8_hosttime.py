"""Width-8 readiness on a ONE-GPU box: W gloo ranks of the REAL candidate evaluator (BASELINE config 5's per-GPU
share: 64 candidates x one eval_cfg episode each), all mapped to device 0, against one rank doing the same.
The GPU is shared, so WALL time grows with W by construction; what the run shows is the HOST side of a rank --
CPU seconds the rank's process consumed (user + system, all threads) per evaluation, and the time it spent
outside the C ABI call -- which must not grow with W if 8 ranks on an 8-GPU node are to scale: every rank's
host work has to fit beside the other seven on the node's cores.
python tools/world8_hosttime.py [world=8] [candidates_per_rank=64] [rows=200]"""
import os
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import numpy as np                                                    # noqa: E402


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, per_rank, rows, q, pin):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import resource
    import torch.distributed as dist
    from autompc_amd.synthetic import make_workload
    from autompc_amd.tuning import CandidateEvaluator, evaluate_sharded, random_candidates
    from autompc_amd.tuning.hostpin import pin_rank
    rec = pin_rank(rank, world) if (pin and world > 1) else None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    system, task, model, _ = make_workload("c3", precision="f64", device=0)
    task.set_num_steps(rows)
    cands = random_candidates(system, per_rank * world, seed=0)
    ev = CandidateEvaluator(system, task, model, device=0)
    inside = [0.0]

    def local(shard, lo):
        t0 = time.perf_counter()
        out = ev.evaluate(shard, seed=0, index_offset=lo)
        inside[0] += time.perf_counter() - t0
        return out
    evaluate_sharded(local, cands, weights="auto")           # untimed pass: plans, JIT lookups
    if world > 1:
        dist.barrier()
    inside[0] = 0.0
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    scores = evaluate_sharded(local, cands, weights="auto")
    wall = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    q.put(dict(rank=rank, wall_s=wall, cpu_s=cpu, evaluate_s=inside[0], outside_s=wall - inside[0],
               cpus=None if rec is None else len(rec["cpus"]), checksum=float(np.nansum(scores[:per_rank]))))
    if world > 1:
        dist.destroy_process_group()


def run(world, per_rank, rows, pin=True):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, per_rank, rows, q, pin)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=3000) for _ in range(world)), key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=300)
    return out


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    per_rank = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    print("host: %d CPUs allowed; every rank on device 0; %d candidates x %d rows per rank" %
          (len(os.sched_getaffinity(0)), per_rank, rows))
    base = None
    for w, pin in ((1, False), (2, True), (world, True), (world, False)):
        res = run(w, per_rank, rows, pin)
        cpu = np.array([d["cpu_s"] for d in res])
        wall = np.array([d["wall_s"] for d in res])
        outside = np.array([d["outside_s"] for d in res])
        if base is None:
            base = dict(cpu=cpu.mean(), wall=wall.mean())
        print("world %d %-8s wall/rank %.3f s (x%.2f of 1 rank; the GPU is shared)   host CPU/rank mean %.3f s max %.3f s   "
              "cores busy per rank %.2f   outside the evaluator (max over ranks) %.4f s   CPUs/rank %s"
              % (w, "pinned" if pin and w > 1 else "unpinned", wall.mean(), wall.mean() / base["wall"], cpu.mean(),
                 cpu.max(), cpu.mean() / wall.mean(), outside.max(), res[0]["cpus"]))
    print("reading: wall time per rank grows with W because W ranks share ONE GPU here.  What a rank asks of the host is "
          "'cores busy per rank' = CPU seconds / wall seconds: the calling thread polls inside the C ABI for the whole "
          "episode (hipStreamSynchronize spins) plus the HIP runtime's helper thread -- if that figure is the same at "
          "W = 1, 2 and 8, a rank's host side does not depend on the world size and an 8-GPU node needs 8 x that many "
          "cores (it has hundreds).  'outside the evaluator' is everything that is not the device episode: sharding, the "
          "gloo all-gather, Python.")
