#!/usr/bin/env python3
"""Headline benchmark: particle-updates/s of the physics step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c1|c0|c3]

N = 1: workload C2 -- 4,194,304 particles, bounds 90000x45000 (the reference's
default density), state drawn exactly like the reference's __init__ with seed
42, every stage on (gravity + collisions + conduction + cooling + walls +
explosions).  N > 1: workload C3 (16,777,216-particle rotating disk), ONE
simulation sharded over the N ranks, one process per GPU (torch.distributed /
RCCL).  `--gpus N` without a launcher's WORLD_SIZE in the environment starts the N rank
processes itself (fresh children, started before anything touches the GPU; rank r on
cuda:r) and passes rank 0's JSON line through; under torchrun (WORLD_SIZE set) the
process is one rank of the launcher's job.

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


def workload(name, n=None):
    """-> (label, n, bounds, state-or-None, stage_mask).  n: other size for the c4 recipe."""
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
    if name == "c4":
        # dense stress case: uniform disk at 4x the default density, at rest, temperatures
        # U[0.5, 0.9] so that collision heating pushes particles past the explosion threshold
        # (64 Mi particles: R = 73078 in 150000 x 150000; other sizes keep the density)
        n = n or (1 << 26)
        radius = float(np.sqrt(n / (np.pi * 4e-3)))
        side = 2.0 * radius * 1.0263
        b = (side, side)
        st = disk_state(n, b, radius, rng)
        st['temperatures'] = rng.uniform(0.5, 0.9, (n, 1))
        return "C4 %d-particle dense hot disk R=%.0f (4x default density), all stages" % (n, radius), n, b, st, 31
    raise SystemExit("unknown workload " + name)


def cpu_baseline(n, bounds, state, primed_accelerations, mask, label):
    """The oracle (C restatement of the reference's single-core CPU path) timed on this
    host on the benchmark's own initial state, full size: ONE update() after priming.
    The priming accelerations a(0) are handed over from the device run (priming is outside
    the timed region on both sides, and computing it again on one core would only add a
    minute of untimed work).  The build used is the one whose six `fastmath` stages of the
    reference (tree walk, potential energy, the two sweeps' sums, conduction) are compiled
    with -ffast-math as Numba would; the strict stages stay strict."""
    from oracle import oracle as orc
    o = orc.OracleSystem(n, SIM, np.random.default_rng(42), bounds, state=state, stage_mask=mask, fast_stages=True)
    o.set_fast_pow(True)      # x**-1.5 as 1/(x sqrt x), what an optimising JIT emits
    o.accelerations[:] = primed_accelerations
    t0 = time.perf_counter()
    o.update()
    dt = time.perf_counter() - t0
    return {"value": o.num_particles / dt, "unit": "particle-updates/s", "cores": 1, "kind": "port",
            "seconds": dt,
            "sample": "1 full step of the workload itself (%s; %d particles) after priming, oracle/psim_oracle.c, "
                      "gcc -O2 with -ffast-math on the reference's fastmath stages, one thread" % (label, n)}


def other_workload(name, steps, warmup, device, cpu_steps=0):
    """One of the other single-GPU configurations, compact: the same timed loop as the main
    measurement (all steps in one call of the library, state resident), the walk's share, and
    for C0 -- the reference's own default run -- the oracle timed on the same state for
    `cpu_steps` steps (prime, then update() that many times, as the reference's main.py does)."""
    from psim_amd.particle_system import ParticleSystem
    label, n, bounds, state, mask = workload(name)
    ps = ParticleSystem(n, dict(SIM), np.random.default_rng(42), bounds, device=device, stage_mask=mask, state=state)
    ps._calculate_gravity()
    ps.update(warmup)
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ps.update(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    updates = ps.particle_updates()
    stage_ms = ps.stage_times_ms()
    visits, inter = ps.walk_counts()
    flops = 20.0 * inter + 8.0 * (visits - inter)
    walk_ms = stage_ms.get("gravity", 0.0)
    rec = {"workload": label, "particles": n, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
           "value": updates / dt, "unit": "particle-updates/s", "walk_kernel_ms": walk_ms,
           "walk_frac_of_fp64_valu_peak": (flops / (walk_ms * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS) if walk_ms > 0 else 0.0,
           "contacts_last_step": ps.counts()["contacts"]}
    ps.close()
    if cpu_steps:
        from oracle import oracle as orc
        o = orc.OracleSystem(n, SIM, np.random.default_rng(42), bounds, state=state, stage_mask=mask, fast_stages=True)
        o.set_fast_pow(True)
        o._calculate_gravity()
        t0 = time.perf_counter()
        done = 0
        for _ in range(cpu_steps):
            o.update()
            done += o.num_particles
        cdt = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": done / cdt, "unit": "particle-updates/s", "cores": 1, "kind": "port", "seconds": cdt,
                               "sample": "%d steps of the workload itself after priming, oracle/psim_oracle.c, gcc -O2 with "
                                         "-ffast-math on the reference's fastmath stages, one thread" % cpu_steps}
    return rec


def launch_ranks(n_ranks):
    """Start one fresh process per rank (this process has not touched the GPU or imported
    torch) and hand rank 0's output through.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if "--one-device" in sys.argv else str(r), WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # all children are watched together: when one fails the others (blocked in the next
    # collective) are ended and its exit code is reported at once
    rc = 0
    deadline = time.time() + float(os.environ.get("PSIM_BENCH_RANKS_TIMEOUT", "3000"))
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-walk-alone", action="store_true",
                    help="do not time the walk kernel with the device to itself after the steps (profiling runs: every "
                         "launch of the kernel in the trace is then one of the steps' own, but for the priming call)")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="the default single-GPU run also measures, compactly, the other single-GPU configurations "
                         "(C0 with its CPU baseline, C1, C3 on one GPU): skip that")
    ap.add_argument("--walk-mode", type=int, default=0)
    ap.add_argument("--library", default=None,
                    help="alternative build of the library (kernel experiments; the test-only host-loop build, "
                         "with --backend gloo, for the CPU test of the multi-rank path)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--one-device", action="store_true",
                    help="with --gpus N --backend gloo: every rank on GPU 0 (a rehearsal of the multi-rank path on a "
                         "one-GPU box; the ranks share the device, so the timings mean nothing)")
    ap.add_argument("--staged", action="store_true",
                    help="drive the step through the psim_stage_* entry points (the multi-GPU path) even on one GPU")
    ap.add_argument("--emulate-rank", type=int, default=None,
                    help="with --staged --world W: ONE process on one GPU plays all W ranks of the sharded step in turn and "
                         "times the share of this rank alone (what one GPU of a W-GPU job computes per step, and the bytes "
                         "it exchanges); the other ranks' shares are computed untimed so that the state stays exact")
    ap.add_argument("--world", type=int, default=8, help="number of ranks played with --emulate-rank")
    ap.add_argument("--no-overlap", action="store_true",
                    help="every stage on one stream (kernel timings without the side stream's contention)")
    ap.add_argument("--brute", type=int, default=262144,
                    help="also time the all-pairs cross-check kernel on this many particles (0 = skip)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")

    # --gpus N without a launcher: this process only starts the N ranks and waits for them
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

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
    on_gpu = args.library is None or args.backend == "nccl" or os.path.basename(args.library).startswith("libpsim_hip")
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP library has no CPU fallback)")
        torch.cuda.set_device(local_rank)
    emulate = None
    if args.emulate_rank is not None:
        if world != 1:
            raise SystemExit("--emulate-rank plays the ranks in one process: use it with --gpus 1")
        emulate = (args.world, args.emulate_rank)
        args.staged = True
    if (world > 1 or args.staged) and emulate is None:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from psim_amd.particle_system import ParticleSystem
    # N = 1: C2 (4Mi, the single-GPU configuration the metric is quoted on).
    # N > 1: C3 (16Mi rotating disk), ONE simulation sharded over the N ranks: every
    # rank holds the state, gravity is split by sorted-target range, collisions and conduction
    # by slabs of grid rows, results all-gathered over RCCL (psim_amd/distributed.py).
    wl = args.workload or ("c2" if world == 1 and emulate is None else "c3")
    label, n, bounds, state, mask = workload(wl)
    if world > 1 or args.staged:
        from psim_amd.distributed import ShardedParticleSystem
        ps = ShardedParticleSystem(n, dict(SIM), np.random.default_rng(42), bounds, device=local_rank,
                                   stage_mask=mask, walk_mode=args.walk_mode, state=state, library=args.library,
                                   emulate=emulate)
    else:
        ps = ParticleSystem(n, dict(SIM), np.random.default_rng(42), bounds, device=local_rank,
                            stage_mask=mask, walk_mode=args.walk_mode, state=state, library=args.library,
                            overlap=not args.no_overlap)
    want_cpu = not args.no_cpu_baseline and world == 1 and rank == 0 and emulate is None
    initial = None
    if want_cpu:
        initial = dict(positions=ps.positions.copy(), masses=ps.masses.copy(), temperatures=ps.temperatures.copy(),
                       velocities=ps.velocities.copy())
    ps._calculate_gravity()
    primed = ps.accelerations.copy() if want_cpu else None
    for _ in range(args.warmup):
        ps.update()
    if world > 1:
        ps.warm_up_exchanges()       # (with --warmup 0 the collectives' first use would be timed)
    if emulate is not None:
        ps.emulated_ms = {}

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    stage_acc = {}
    steps = max(args.steps, 1)
    barrier()
    t0 = time.perf_counter()
    updates = 0
    if world == 1 and not args.staged:
        # all K steps in one call of the library: nothing between two steps but the next
        # step's launches; the stage times are averages the library takes from HIP events it
        # recorded along the way, read after the call
        ps.update(args.steps)
        barrier()
        dt = time.perf_counter() - t0
        updates = ps.particle_updates()
        stage_ms = ps.stage_times_ms()
    else:
        for _ in range(args.steps):
            ps.update()
            updates += ps.num_particles
            for k, v in ps.stage_times_ms().items():
                stage_acc[k] = stage_acc.get(k, 0.0) + v
        barrier()
        dt = time.perf_counter() - t0
        stage_ms = {k: v / steps for k, v in stage_acc.items()}
    emulated = None
    if emulate is not None:
        # the rank's own share: the sum of its timed pieces (host clock, device idle at both
        # ends of every piece); the other ranks' pieces, computed in between, are not in it
        pieces = {k: v / steps for k, v in ps.emulated_ms.items()}
        dt = sum(ps.emulated_ms.values()) * 1e-3
        link_gbs = 153.0
        xb = dict(ps.exchange_bytes)
        W = emulate[0]
        # all-gather over the 7 direct xGMI links of a GPU: every rank receives W - 1 shards, one per link
        xb["accelerations_received"] = xb["accelerations_per_rank"] * (W - 1)
        xb["blocks_received"] = xb["block_sent_per_rank"] * (W - 1)
        emulated = {"world": W, "rank": emulate[1], "ms_per_step_by_piece": pieces, "exchange_bytes": xb,
                    "exchange_ms_at_link_rate": {"accelerations": xb["accelerations_per_rank"] / (link_gbs * 1e6),
                                                 "blocks": xb["block_sent_per_rank"] / (link_gbs * 1e6),
                                                 "assumed_GB_per_s_per_link": link_gbs},
                    "rows": ps._row_bounds, "fallbacks": ps.fallbacks,
                    "note": "ms_per_step and value are this rank's compute share only (exchanges excluded)"}
    walk_ms_ranks = [stage_ms.get("gravity", 0.0)]
    if world > 1:
        # one simulation: every rank counted the same updates; time = slowest rank
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
        every = [None] * world
        dist.all_gather_object(every, stage_ms.get("gravity", 0.0))
        walk_ms_ranks = [float(v) for v in every]

    if rank == 0:
        # dominant kernel: the Barnes-Hut walk (one launch per step).  Algorithmic
        # flops per launch = 20 I + 8 (V - I), I = accepted interactions, V = node
        # visits summed over particles, both counted on the current tree.
        visits, inter = ps.walk_counts()
        flops = 20.0 * inter + 8.0 * (visits - inter)
        if world > 1 or emulate is not None:
            # a rank's launch walks 1/world of the (Morton-sorted) targets: per-GPU figure,
            # taking the ranks' shares of the work as equal
            flops /= (emulate[0] if emulate is not None else world)
        walk_ms = max(walk_ms_ranks)
        achieved = flops / (walk_ms * 1e-3) / 1e12 if walk_ms > 0 else 0.0
        counts = ps.counts()
        # In the step the walk shares the device: it holds part of the wave slots and the
        # grid build and contact solver run in the rest.  The same kernel on the same tree
        # with the device to itself, for comparison:
        alone_ms = ps.time_walk(3, 0.0) if (world == 1 and on_gpu and emulate is None and not args.no_walk_alone) else None
        out = {
            "metric": "particle-updates/sec", "value": updates / dt, "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": label, "particles": n, "theta": SIM["barnes_hut_theta"],
                       "multi_gpu": ("one simulation sharded over %d ranks: walk by sorted-target range, collisions and conduction "
                                     "by slabs of grid rows (own rows + halo, seam records verified), RCCL all-gathers" % world) if world > 1 else
                                     ("rank %d of %d played by one process (its compute share only)" % (emulate[1], emulate[0])
                                      if emulate is not None else
                                      "single (stage entry points of the sharded path, one rank)" if args.staged else "single")},
            "roofline": {"kernel": "gravity_walk_multi_kernel", "bound": "valu", "achieved": achieved,
                         "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_VALU_TFLOPS, "traffic": None,
                         "kernel_ms": walk_ms, "interactions_per_particle": inter / max(ps.num_particles, 1),
                         "visits_per_particle": visits / max(ps.num_particles, 1),
                         "wave_trips_per_wave": ps.last_walk_trips[0] / max((ps.num_particles + 63) // 64, 1),
                         "force_trips_per_wave": ps.last_walk_trips[1] / max((ps.num_particles + 63) // 64, 1),
                         "note": "kernel_ms is the walk's duration inside the timed steps (HIP events on the stream it is "
                                 "launched on), where other stages run concurrently in wave slots the walk leaves free; "
                                 "device_to_itself is the same launch with nothing else running.  traffic: the walk is "
                                 "issue-bound, its HBM counters are kept under profiles/ and not repeated here"},
            "stage_ms": stage_ms,
            "last_step": {"contacts": counts["contacts"], "solver_iterations": counts["solver_iterations"],
                          "solver_rounds": counts["solver_rounds"], "tree_nodes": counts["tree_nodes"],
                          "particles": counts["particles"]},
        }
        if emulated is not None:
            out["emulated_rank"] = emulated
            out["roofline"]["note"] = "walk kernel time is not separated in this mode (see ms_per_step_by_piece)"
        if world > 1:
            out["sharding"] = {"rows": ps._row_bounds, "halo_rows": ps._halo, "whole_grid_fallbacks": ps.fallbacks,
                               "exchange_bytes": ps.exchange_bytes}
            out["roofline"]["kernel_ms_per_rank"] = walk_ms_ranks
            out["roofline"]["frac_per_rank"] = [flops / (m * 1e-3) / 1e12 / PEAK_FP64_VALU_TFLOPS if m > 0 else 0.0
                                                for m in walk_ms_ranks]
        if alone_ms:
            tf = flops / (alone_ms * 1e-3) / 1e12
            out["roofline"]["device_to_itself"] = {"kernel_ms": alone_ms, "achieved": tf, "frac": tf / PEAK_FP64_VALU_TFLOPS}
        if args.brute and on_gpu and world == 1 and emulate is None:
            # yardstick: the same force arithmetic with no tree and no divergence
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
        if (world == 1 and on_gpu and emulate is None and not args.staged and args.workload is None and
                args.library is None and not args.no_other_workloads):
            # the other single-GPU configurations, so that the one line holds them all
            ps.close()
            out["other_workloads"] = {
                "c0": other_workload("c0", 100, 10, local_rank, cpu_steps=0 if args.no_cpu_baseline else 100),
                "c1": other_workload("c1", 20, 3, local_rank),
                "c3_one_gpu": other_workload("c3", 10, 2, local_rank)}
        if want_cpu:      # (one rank only: the other ranks would wait at the barrier)
            ps.close()
            out["cpu_baseline"] = cpu_baseline(n, bounds, initial, primed, mask, label)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if (world > 1 or args.staged) and emulate is None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
