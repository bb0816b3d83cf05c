"""The fan-out launch's work-item order (csrc/conv_igemm.hip, conv_igemm_kernel: ConvArgs.fan_chunk, round 6): inside the contiguous
index space the XCD remap leaves, work item b -> (head, pixel tile) runs in chunks of T tiles, head-major inside a chunk, the last chunk
short.  Host restatement of the kernel's index arithmetic: every (head, tile) pair must be produced exactly once for every launch
shape -- an index that is produced twice is a tile computed twice and another one never written."""
import itertools

import pytest


def _item(b, nx, groups, chunk):
    """(head, tile) of work item b -- the kernel's arithmetic, integer for integer."""
    if chunk <= 0:
        return b % groups, b // groups
    per = chunk * groups
    c = b // per
    w = b - c * per
    here = min(chunk, nx - c * chunk)
    head = w // here
    return head, c * chunk + (w - head * here)


@pytest.mark.parametrize("groups", [2, 3])
@pytest.mark.parametrize("chunk", [0, 1, 8, 16, 64, 4096])
def test_every_head_and_tile_exactly_once(groups, chunk):
    for nx in (1, 2, 7, 15, 16, 17, 31, 32, 33, 341, 1364, 10912):
        seen = [_item(b, nx, groups, chunk) for b in range(nx * groups)]
        assert len(set(seen)) == nx * groups, (nx, groups, chunk)
        assert set(seen) == set(itertools.product(range(groups), range(nx))), (nx, groups, chunk)


def test_chunks_are_head_major_and_tiles_stay_together():
    """A chunk's tiles are visited by head 0, then the same tiles by head 1, ...: consecutive work items of one head are consecutive tiles."""
    nx, groups, chunk = 37, 3, 16
    seen = [_item(b, nx, groups, chunk) for b in range(nx * groups)]
    assert seen[:16] == [(0, t) for t in range(16)] and seen[16:32] == [(1, t) for t in range(16)] and seen[32:48] == [(2, t) for t in range(16)]
    assert seen[-5:] == [(2, t) for t in range(32, 37)]          # the short last chunk: 5 tiles per head
    assert _item(5, nx, groups, 0) == (2, 1)                     # interleaved order: tile 1, head 2
