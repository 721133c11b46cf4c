"""Host-side thread policy (agile3d_amd/hostcpu.py): torch's CPU pool follows the container's CPU quota."""
import os
import subprocess
import sys

import torch

from agile3d_amd import hostcpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quota_is_positive_and_within_the_affinity_mask():
    q = hostcpu.cpu_quota()
    assert 1.0 <= q <= len(os.sched_getaffinity(0))


def test_import_caps_the_pool_to_half_the_quota():
    want = max(1, int(hostcpu.cpu_quota() // 2))
    assert torch.get_num_threads() <= max(want, 1)
    assert hostcpu.cap_host_threads() == torch.get_num_threads()


def _threads_in_child(env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, "-c", "import torch, agile3d_amd; print(torch.get_num_threads())"], cwd=ROOT, env=e,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return int(out.stdout.strip().splitlines()[-1])


def test_switch_forces_a_count_or_leaves_torch_alone():
    assert _threads_in_child({"A3D_HOST_THREADS": "3"}) == 3
    assert _threads_in_child({"A3D_HOST_THREADS": "0", "OMP_NUM_THREADS": "5"}) == 5
    # OMP_NUM_THREADS below the cap is respected (the cap never raises the count)
    assert _threads_in_child({"OMP_NUM_THREADS": "1"}) == 1
    # the ranks of a node share the quota
    want = max(1, int(hostcpu.cpu_quota() // (2 * 4)))
    assert _threads_in_child({"LOCAL_WORLD_SIZE": "4"}) <= want


def test_row_blocks_back_to_back_are_taken_as_a_view():
    """train_decoder._rows_as_one: the samples' row blocks of a batch (slices of the backbone's output, of the batch's position
    encodings) become ONE [N_total, C] tensor without a copy when they already lie back to back in one storage; anything else
    (a gap, another storage, another dtype or width, a non-contiguous block) is concatenated."""
    import torch
    from agile3d_amd.train_decoder import _rows_as_one
    a = torch.arange(40.).view(10, 4)
    v = _rows_as_one([a[0:3], a[3:7], a[7:10]])
    assert v.data_ptr() == a.data_ptr() and v.shape == (10, 4) and torch.equal(v, a)
    w = _rows_as_one([a[2:5], a[5:9]])                       # a window that does not start at the storage's first row
    assert w.data_ptr() == a[2:].data_ptr() and torch.equal(w, a[2:9])
    for blocks in ([a[0:3], a[4:7]],                           # a gap
                   [a[0:3], torch.ones(3, 4)],                 # another storage
                   [a[0:3], a[3:7].double()],                  # another dtype
                   [a[0:3, :2], a[3:7, :2]]):                  # non-contiguous blocks
        c = _rows_as_one(blocks)
        assert c.data_ptr() != a.data_ptr() and torch.equal(c.double(), torch.cat([b.double() for b in blocks], 0))
    one = _rows_as_one([a[1:4]])
    assert one.data_ptr() == a[1:].data_ptr()
