#!/usr/bin/env python3
"""Headline benchmark: particle-updates/s of the physics step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c0|c3]

N = 1: workload C2 -- 4,194,304 particles, bounds 90000x45000 (the reference's
default density), state drawn exactly like the reference's __init__ with seed
42, every stage on (gravity + collisions + conduction + cooling + walls +
explosions).  N > 1: workload C3 (16,777,216-particle rotating disk), ONE
simulation sharded over the N ranks, one process per GPU (torch.distributed /
RCCL).

A "step" is one ParticleSystem.update().  The timed region starts with the
state resident in HBM; value = sum over timed steps of alive particles / wall
time (max over ranks).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "particle-simulator_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np

SIM = dict(particle_count=2000, min_mass=5.0, max_mass=20.0, gravity_constant=0.01, softening_factor=10,
           barnes_hut_theta=0.5, min_temp=0.05, max_temp=0.1, heat_transfer_coefficient=0.1,
           coefficient_of_restitution=0.5, grid_cell_size_multiplier=10.0, position_correction_factor=1.05,
           collision_solver_iterations=1, explosion_efficiency=0.5, thermal_damping_factor=0.0005)

# MI355X peaks used for the roofline fraction (spec values)
PEAK_FP64_VALU_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def disk_state(n, bounds, radius, rng, rotating=False, g_const=0.01):
    """uniform disk about the centre: r = R sqrt(u1), phi = 2 pi u2 (draw order u1, u2, mass, temp)"""
    u1 = rng.random(n)
    u2 = rng.random(n)
    masses = rng.uniform(SIM['min_mass'], SIM['max_mass'], (n, 1))
    temps = rng.uniform(SIM['min_temp'], SIM['max_temp'], (n, 1))
    r = radius * np.sqrt(u1)
    phi = 2 * np.pi * u2
    pos = np.stack([bounds[0] / 2 + r * np.cos(phi), bounds[1] / 2 + r * np.sin(phi)], axis=1)
    st = dict(positions=pos, masses=masses, temperatures=temps)
    if rotating:
        vc = np.sqrt(g_const * masses.sum() * r / radius ** 2)
        st['velocities'] = np.stack([-vc * np.sin(phi), vc * np.cos(phi)], axis=1)
    return st


def workload(name):
    """-> (label, n, bounds, state-or-None, stage_mask, seed)"""
    rng = np.random.default_rng(42)
    if name == "c0":
        return "C0 default config.json: 2000 particles, 2000x1000, all stages", 2000, (2000, 1000), None, 31
    if name == "c1":
        n, b = 1 << 20, (40000, 40000)
        return ("C1 1Mi uniform disk R=18269, gravity + integration only", n, b,
                disk_state(n, b, 18269.0, rng), 0)
    if name == "c2":
        return ("C2 4Mi particles, 90000x45000, reference init (seed 42), all stages",
                1 << 22, (90000, 45000), None, 31)
    if name == "c3":
        n, b = 1 << 24, (150000, 150000)
        return ("C3 16Mi rotating disk R=73078, all stages", n, b,
                disk_state(n, b, 73078.0, rng, rotating=True), 31)
    raise SystemExit("unknown workload " + name)


def cpu_baseline(seconds_budget=20.0):
    """The oracle (C restatement of the reference's single-core CPU path) timed
    on this host on a bounded sample of the C2 workload: 262,144 particles at
    the same density (bounds 22500x11250), all stages."""
    from oracle import oracle as orc
    n, bounds = 1 << 18, (22500, 11250)
    o = orc.OracleSystem(n, SIM, np.random.default_rng(42), bounds)
    o.set_fast_pow(True)      # x**-1.5 as 1/(x sqrt x), what an optimising JIT emits
    o._calculate_gravity()
    t0 = time.perf_counter()
    steps = 0
    updates = 0
    while steps < 2 or (time.perf_counter() - t0 < seconds_budget and steps < 50):
        o.update()
        steps += 1
        updates += o.num_particles
    dt = time.perf_counter() - t0
    return {"value": updates / dt, "unit": "particle-updates/s", "cores": 1, "kind": "port",
            "sample": "%d steps of a %d-particle cut of the workload at the same density (bounds %dx%d), "
                      "oracle/psim_oracle.c, gcc -O2, one thread" % (steps, n, bounds[0], bounds[1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--walk-mode", type=int, default=0)
    ap.add_argument("--library", default=None, help="alternative build of libpsim_hip.so (kernel experiments)")
    ap.add_argument("--staged", action="store_true",
                    help="drive the step through the psim_stage_* entry points (the multi-GPU path) even on one GPU")
    ap.add_argument("--no-overlap", action="store_true",
                    help="every stage on one stream (kernel timings without the side stream's contention)")
    ap.add_argument("--brute", type=int, default=0,
                    help="also time the all-pairs cross-check kernel on this many particles")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: libraries that print there (RCCL announces its
    # version on stdout when a communicator is created) are sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.staged:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from psim_amd.particle_system import ParticleSystem
    # N = 1: C2 (4Mi, the single-GPU configuration the metric is quoted on).
    # N > 1: C3 (16Mi rotating disk), ONE simulation sharded over the N ranks: every
    # rank holds the state, gravity is split by sorted-target range, contact detection and
    # conduction by grid-slot range, outputs all-gathered over RCCL (psim_amd/distributed.py).
    wl = args.workload or ("c2" if world == 1 else "c3")
    label, n, bounds, state, mask = workload(wl)
    if world > 1 or args.staged:
        from psim_amd.distributed import ShardedParticleSystem
        ps = ShardedParticleSystem(n, dict(SIM), np.random.default_rng(42), bounds, device=local_rank,
                                   stage_mask=mask, walk_mode=args.walk_mode, state=state)
    else:
        ps = ParticleSystem(n, dict(SIM), np.random.default_rng(42), bounds, device=local_rank,
                            stage_mask=mask, walk_mode=args.walk_mode, state=state, library=args.library,
                            overlap=not args.no_overlap)
    ps._calculate_gravity()
    for _ in range(args.warmup):
        ps.update()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stage_acc = {}
    barrier()
    t0 = time.perf_counter()
    updates = 0
    for _ in range(args.steps):
        ps.update()
        updates += ps.num_particles
        for k, v in ps.stage_times_ms().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        # one simulation: every rank counted the same updates; time = slowest rank
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])

    if rank == 0:
        steps = max(args.steps, 1)
        stage_ms = {k: v / steps for k, v in stage_acc.items()}
        # dominant kernel: the Barnes-Hut walk (one launch per step).  Algorithmic
        # flops per launch = 20 I + 8 (V - I), I = accepted interactions, V = node
        # visits summed over particles, both counted on the current tree.
        visits, inter = ps.walk_counts()
        flops = 20.0 * inter + 8.0 * (visits - inter)
        if world > 1:
            # a rank's launch walks 1/world of the (Morton-sorted) targets: per-GPU figure,
            # taking the ranks' shares of the work as equal
            flops /= world
        walk_ms = stage_ms.get("gravity", 0.0)
        achieved = flops / (walk_ms * 1e-3) / 1e12 if walk_ms > 0 else 0.0
        counts = ps.counts()
        # In the step the walk shares the device: it holds part of the wave slots and the
        # grid build and contact solver run in the rest.  The same kernel on the same tree
        # with the device to itself, for comparison:
        alone_ms = ps.time_walk(3, 0.0) if world == 1 else None
        out = {
            "metric": "particle-updates/sec", "value": updates / dt, "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": label, "particles": n, "theta": SIM["barnes_hut_theta"],
                       "multi_gpu": ("one simulation sharded over %d ranks: walk by sorted-target range, contact detection "
                                     "and conduction by grid-slot slabs, RCCL all-gathers" % world) if world > 1 else "single"},
            "roofline": {"kernel": "gravity_walk_multi_kernel", "bound": "valu", "achieved": achieved,
                         "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": None,
                         "kernel_ms": walk_ms, "interactions_per_particle": inter / max(ps.num_particles, 1),
                         "visits_per_particle": visits / max(ps.num_particles, 1),
                         "wave_trips_per_wave": ps.last_walk_trips[0] / max((ps.num_particles + 63) // 64, 1),
                         "force_trips_per_wave": ps.last_walk_trips[1] / max((ps.num_particles + 63) // 64, 1),
                         "note": "kernel_ms is the walk's duration inside the timed steps, where other stages run "
                                 "concurrently in wave slots the walk leaves free; device_to_itself is the same "
                                 "launch with nothing else running"},
            "stage_ms": stage_ms,
            "last_step": {"contacts": counts["contacts"], "solver_iterations": counts["solver_iterations"],
                          "solver_rounds": counts["solver_rounds"], "tree_nodes": counts["tree_nodes"],
                          "particles": counts["particles"]},
        }
        if wl == "c2" and world == 1:
            # recorded, not measured by this run: counters need rocprofv3 passes of their own
            out["roofline"]["traffic_recorded"] = {
                "fetch_bytes": 2.69e9, "write_bytes": 1.35e8, "algorithmic_bytes": 4.4e8,
                "source": "profiles/r1_walk_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                          "raw counters; 32-byte scalar fetches are not a calibrated pattern on gfx950)"}
        if alone_ms:
            tf = flops / (alone_ms * 1e-3) / 1e12
            out["roofline"]["device_to_itself"] = {"kernel_ms": alone_ms, "achieved": tf, "frac": tf / PEAK_FP64_VALU_TFLOPS}
        if args.brute:
            nb = args.brute
            side = float(np.sqrt(nb / 1e-3))
            bs = ParticleSystem(nb, dict(SIM), np.random.default_rng(7), (side, side), device=local_rank)
            bs.brute_force_accelerations()
            ms = min(bs.brute_force_accelerations()[1] for _ in range(3))
            tf = 20.0 * nb * (nb - 1) / (ms * 1e-3) / 1e12
            out["brute_force"] = {"kernel": "brute_force_kernel", "particles": nb, "kernel_ms": ms, "bound": "valu",
                                  "achieved": tf, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                                  "frac": tf / PEAK_FP64_VALU_TFLOPS}
            bs.close()
        if not args.no_cpu_baseline and world == 1:      # (one rank only: the other ranks would wait at the barrier)
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.staged:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
