"""world_size 2 with the HIP kernels on a one-GPU box: two rank processes share the GPU.  RCCL
refuses two ranks on one device, so the exchange goes the way `ShardedParticleSystem` takes
whenever the backend cannot move device tensors (transport "host"): a rank's shard is copied
to page-locked host memory on the library's stream, gloo all-gathers the CPU tensors, and the
result is copied back.  Each rank first runs the unsharded step alone and keeps the states,
then the sharded one, so that a process holds one context at a time."""
import pytest

from test_distributed_gloo import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,steps", [("dense256", 6), ("hot300", 6), ("c4:300000", 3)])
def test_sharded_over_two_ranks_on_the_hip_library(name, steps):
    """both ranks check, step by step, that the sharded state is bit-identical to the
    unsharded run: target ranges of the walk that start inside the sorted array, bands of grid
    rows with a seam in the middle of the domain, export blocks regrown on both ranks
    (c4: the dense hot disk, 300 000 particles, contacts in both bands and explosions)"""
    run_ranks(2, "gloo", "hip", name, steps, timeout=150)
