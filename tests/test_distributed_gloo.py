"""The N > 1 path on CPU: world_size 2 and 3, gloo backend, the test-only
host-loop build of the library.  One simulation sharded over the ranks must be
bit-identical to the unsharded run (every element is computed by exactly one
rank with the same arithmetic)."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HOSTLOOP_DIR = os.path.join(HERE, "hostloop")
HOSTLOOP = os.path.join(HOSTLOOP_DIR, "libpsim_hostloop.so")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(world, backend, library, name, steps, timeout=240):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   DIST_WORKER_DUMP_AFTER=str(max(10, int(timeout) - 20)))     # (a stuck rank says where before it is killed)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), backend, library,
                                       name, str(steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert "ok:" in o


@pytest.mark.parametrize("world,name,steps", [(2, "dense256", 6), (2, "hot300", 6), (3, "fast200", 5),
                                              (2, "dense400", 5), (3, "default2000", 4), (4, "c4:60000", 3)])
def test_sharded_equals_unsharded(world, name, steps):
    subprocess.check_call(["make", "-s", "-C", HOSTLOOP_DIR])
    run_ranks(world, "gloo", HOSTLOOP, name, steps)


def test_bench_gpus_2_starts_two_ranks_and_prints_one_line():
    """`bench.py --gpus 2` with no launcher in the environment starts its two ranks itself
    (here: the host-loop build over gloo, the default 2000-particle workload) and rank 0
    prints exactly one JSON line with n_gpus == 2"""
    import json
    subprocess.check_call(["make", "-s", "-C", HOSTLOOP_DIR])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "3",
                          "--warmup", "1", "--workload", "c0", "--library", HOSTLOOP, "--backend", "gloo",
                          "--walk-mode", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "strong"
    assert rec["value"] > 0 and len(rec["roofline"]["kernel_ms_per_rank"]) == 2
    assert "cpu_baseline" not in rec
