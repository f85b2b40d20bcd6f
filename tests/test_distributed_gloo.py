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
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
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
                                              (2, "dense400", 5), (3, "default2000", 4)])
def test_sharded_equals_unsharded(world, name, steps):
    subprocess.check_call(["make", "-s", "-C", HOSTLOOP_DIR])
    run_ranks(world, "gloo", HOSTLOOP, name, steps)
