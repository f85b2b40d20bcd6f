"""world_size 2 with the HIP kernels.  The box has one GPU, so the two ranks share it and
gloo moves the device tensors (RCCL refuses two ranks on one device).  Collected late (before the
allocation-free entry point's test only): several processes on one GPU is not how the library is
deployed.  Three and four ranks run the same way (tests/dist_worker.py under WORLD_SIZE 3 or 4)
and give the same bit-identical states when they finish, but gloo's all-gather of device tensors
then sometimes does not return for minutes (about every third run; the ranks' kernels have
long finished), so those are not in the suite."""
import pytest

from test_distributed_gloo import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,steps", [("dense256", 6), ("hot300", 6), ("c4:300000", 3)])
def test_sharded_over_two_ranks_on_the_hip_library(name, steps):
    """both ranks check, step by step, that the sharded state is bit-identical to their own
    unsharded run: target ranges of the walk that start inside the sorted array, detection
    slabs that start inside the grid, key lists gathered from two ranks (c4: the dense hot
    disk, 300 000 particles, contacts in both slabs and explosions)"""
    run_ranks(2, "gloo", "hip", name, steps, timeout=150)
