"""The one-shot all-reduce as a testable component: W stand-in ranks inside ONE process on one GPU, each with its own
payload / staging rows / flags and its own stream.  The result must be BITWISE the rank-ordered sum on every rank (chunk r is reduced by
rank r alone, p = 0 .. W - 1 in order), call after call (the flags carry a growing sequence number and are never reset), and a rank that
never shows up must end in a timeout status, not in a hung device."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rank_ordered_sum(data):
    acc = data[0].clone()
    for d in data[1:]:
        acc = acc + d          # fp32, one addition per rank, in rank order: what the owner of every chunk computes
    return acc


@pytest.mark.parametrize("world", [2, 4, 8])
def test_single_process_ranks_agree_bitwise_with_the_rank_ordered_sum(world):
    from geometry_rl_amd import oneshot
    n = 135440 + 28 * world + (-(135440 + 28 * world)) % 4      # the actor's gradient slice + the ranks' loss records (agent.PolicyUpdater)
    ranks = oneshot.local_ranks(world, n, DEV)
    g = torch.Generator(device="cpu").manual_seed(world)
    for call in range(3):                                         # three calls back to back on the same areas: the sequence numbers do their job
        data = [torch.randn(n, generator=g).mul_(10.0 ** (r % 3 - 1)).to(DEV) for r in range(world)]
        for r, d in zip(ranks, data):
            r.payload.copy_(d)
        status = oneshot.all_reduce_local(ranks)                  # all ranks' workgroups in one launch: resident together by construction
        torch.cuda.synchronize()
        assert status.tolist() == [0] * world, status.tolist()
        want = _rank_ordered_sum(data)
        for r in ranks:
            assert torch.equal(r.payload, want), (world, call, r.rank, (r.payload - want).abs().max().item())


def test_two_ranks_as_two_launches_on_two_streams():
    """The form a real node runs -- one launch per rank, meeting on the device -- with two stand-in ranks on two streams of this process."""
    from geometry_rl_amd import oneshot
    n = 4096 * 33
    ranks = oneshot.local_ranks(2, n, DEV)
    # two streams of DIFFERENT priority: distinct hardware queues for certain (two streams of one priority may share a queue in this process,
    # and launches that wait for each other on one queue would only meet their timeouts)
    lo, hi = max(torch.cuda.Stream.priority_range()), min(torch.cuda.Stream.priority_range())
    streams = [torch.cuda.Stream(priority=hi), torch.cuda.Stream(priority=lo)]
    g = torch.Generator(device="cpu").manual_seed(1)
    for call in range(3):
        data = [torch.randn(n, generator=g).to(DEV) for _ in range(2)]
        for r, d in zip(ranks, data):
            r.payload.copy_(d)
        torch.cuda.synchronize()
        for r, s in zip(ranks, streams):
            with torch.cuda.stream(s):
                r.all_reduce()
        torch.cuda.synchronize()
        want = _rank_ordered_sum(data)
        for r in ranks:
            r.check()
            assert torch.equal(r.payload, want), (call, r.rank)


def test_a_missing_rank_times_out_instead_of_hanging():
    from geometry_rl_amd import oneshot
    ranks = oneshot.local_ranks(2, 4096, DEV, timeout_ms=200)
    ranks[0].payload.fill_(1.0)
    ranks[0].all_reduce()                                         # rank 1 never enqueues its side
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="timed out"):
        ranks[0].check()
    # the late rank finds rank 0's contribution (flags are sticky: seq only grows) but never its result chunk: a timeout as well, no hang
    ranks[1].payload.fill_(2.0)
    ranks[1].all_reduce()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="a result chunk"):
        ranks[1].check()
    # ... and the group recovers with the next call (a new sequence number on both sides; both stand-in ranks in one launch)
    ranks[0].payload.fill_(1.0)
    ranks[1].payload.fill_(2.0)
    status = oneshot.all_reduce_local(ranks)
    torch.cuda.synchronize()
    assert status.tolist() == [0, 0]
    for r in ranks:
        assert torch.equal(r.payload, torch.full((4096,), 3.0, device=DEV))
