"""world_size > 1 with the HIP kernels.  The box has one GPU, so the ranks share it and gloo
moves the device tensors (RCCL refuses two ranks on one device).  Collected late (before the
allocation-free entry point's test only): several processes with persistent kernels on one
GPU is not how the library is deployed, and a stall here should not hide other tests."""
import pytest

from test_distributed_gloo import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,name,steps", [(2, "dense256", 6), (3, "hot300", 6), (2, "c4:300000", 3)])
def test_sharded_over_several_ranks_on_the_hip_library(world, name, steps):
    """every rank checks, step by step, that the sharded state is bit-identical to its own
    unsharded run: target ranges of the walk that start inside the sorted array, detection
    slabs that start inside the grid, key lists gathered from several ranks (c4: the dense
    hot disk, 300 000 particles, contacts in every slab and explosions)"""
    run_ranks(world, "gloo", "hip", name, steps, timeout=150)
